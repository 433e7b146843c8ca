"""Re-export of the reference config values (see deepinteraction_amd/configs.py)."""
from deepinteraction_amd.configs import POINT_CLOUD_RANGE, decoder_cfg, encoder_pp_cfg  # noqa: F401
