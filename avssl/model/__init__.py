from speechclip_amd.model import *  # noqa: F401,F403
from speechclip_amd.model.kwClip import KWClip_GeneralTransformer  # noqa: F401
