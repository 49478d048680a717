from speechclip_amd.util import add_general_arguments, freeze_model, get_keypadding_mask, unfreeze_model  # noqa: F401
