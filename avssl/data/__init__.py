from speechclip_amd.data import collate_general, collate_to_device  # noqa: F401
