from .attention_modules import VideoAttention, FrameAttention  # noqa: F401
