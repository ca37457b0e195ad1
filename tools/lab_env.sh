# Sourced by the A/B and marginal-cost scripts: they switch developer knobs (RDM_DUP, RDM_GEMM_TUNE, RDM_FUSED_KPCONV=0, ...)
# that exist in the LAB build of the library only (`make -C rdmnet_amd/csrc lab`; common.h: dev_knob).
export RDM_LIB_PATH="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/rdmnet_amd/librdmnet_hip_lab.so"
[ -f "$RDM_LIB_PATH" ] || make -C "$(dirname "$RDM_LIB_PATH")/csrc" lab -j8 > /dev/null
