from speechclip_amd.task import TrainKWClip_GeneralTransformer  # noqa: F401
