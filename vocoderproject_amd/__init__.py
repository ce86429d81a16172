"""vocoderproject_amd -- MI355X-native batch implementation of the DamRsn/VocoderProject DSP hot path.

Product code: csrc/ (HIP kernels + C ABI, built into libvp_amd.so), processor.py (host mirror of the
reference's plugin surface), synth.py (synthetic streams), dist.py (stream sharding across ranks),
offline.py (WAV files in, WAV files out: the notebook's whole-recording flows on the plugin path).
Nothing here imports oracle/.
"""
from .processor import BatchVocoderProcessor, StftRoundTrip, VpError, load_library, PARAM_IDS, KEYS  # noqa: F401
