"""mate_amd: MI355X-native batched MultiAgentTracking step engine (drop-in for the step path of
XuehaiPan/mate).  `import mate_amd` never touches the GPU; the HIP engine is loaded on first use
and there is no CPU fallback."""
from mate_amd.config import ASSETS_DIR, DEFAULT_CONFIG_FILE, read_config, validate_config  # noqa: F401

__version__ = '0.1.0'
