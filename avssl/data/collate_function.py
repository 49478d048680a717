from speechclip_amd.data.collate_function import collate_general, collate_to_device  # noqa: F401
