"""csm-hf_amd: MI355X-native CSM generation path (CSMModel.generate / generate_frame) behind the
reference's Python API.  Host side: Python + torch (device memory, streams, torch.distributed);
compute: hand-written HIP kernels for gfx950 behind the C-ABI in `include/csm_hip.h`."""
from .configuration_csm import CSMConfig, LlamaSubConfig  # noqa: F401
from .modeling_csm import CSMModel, CSMOutput, CSMKVCache, sample_topk  # noqa: F401
from .processor import CSMProcessor  # noqa: F401
from .serving import ContinuousBatcher  # noqa: F401
from .mimi import MimiDecoder, MimiDecodeConfig  # noqa: F401

__all__ = ["CSMConfig", "LlamaSubConfig", "CSMModel", "CSMOutput", "CSMKVCache", "sample_topk", "CSMProcessor", "ContinuousBatcher", "MimiDecoder", "MimiDecodeConfig"]
