"""rdmnet_amd -- MI355X (gfx950) implementation of RDMNet's dense-matching inference path.

Host-side mirror of the reference's operator API on top of the C-ABI library
`librdmnet_hip.so` (sources in rdmnet_amd/csrc, interface in include/rdmnet_hip.h).
PyTorch is used for device memory and streams only.
"""
import os as _os

# Several pairs in flight per GPU (rdmnet_amd.pipeline): one hardware queue per HIP stream.  The runtime reads this when the
# process makes its first HIP call, so it is set at import unless the caller chose a value (docs/EXPERIMENTS.md 5b: four worker streams
# + the default stream on the runtime's default of four queues run at 345 instead of 460 pairs/s).
if 'GPU_MAX_HW_QUEUES' not in _os.environ:
    import sys as _sys
    _torch = _sys.modules.get('torch')
    if _torch is not None and _torch.cuda.is_initialized():
        # (too late -- the runtime read its environment at its first call; say so instead of setting it silently)
        import warnings as _warnings
        _warnings.warn('rdmnet_amd: HIP was initialised before GPU_MAX_HW_QUEUES could be set; more than three pairs in '
                       'flight per GPU will share hardware queues (export GPU_MAX_HW_QUEUES=8 before the first HIP call)',
                       RuntimeWarning, stacklevel=2)
    else:
        _os.environ['GPU_MAX_HW_QUEUES'] = '8'

__version__ = '0.2.0'
