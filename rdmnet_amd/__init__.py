"""rdmnet_amd -- MI355X (gfx950) implementation of RDMNet's dense-matching inference path.

Host-side mirror of the reference's operator API on top of the C-ABI library
`librdmnet_hip.so` (sources in rdmnet_amd/csrc, interface in include/rdmnet_hip.h).
PyTorch is used for device memory and streams only.
"""
__version__ = '0.1.0'
