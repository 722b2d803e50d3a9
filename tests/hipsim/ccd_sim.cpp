// ccd_sim.cpp - builds the product's kernels + C ABI against the CPU SIMT executor (test infrastructure).
#include "hipsim.h"
#include "../../ccd_amd/csrc/abi_impl.h"
