// ccd_hip.hip - the single translation unit of libccd_hip.so (hipcc --offload-arch=gfx950).
#include "prelude_hip.h"
#include "abi_impl.h"
