"""speechclip_amd -- MI355X-native SpeechCLIP forward/contrastive hot path (HIP kernels behind a C ABI)."""
__version__ = "0.1.0"
