from speechclip_amd.optim import get_scheduler  # noqa: F401
