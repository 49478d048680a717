from .kwClip import *  # noqa: F401,F403
