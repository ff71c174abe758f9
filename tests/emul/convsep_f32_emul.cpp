// TEST INFRASTRUCTURE: libvips_amd/csrc/convsep_f32.hip ITSELF (the round-1 separable kernel: every format and
// mask convsep_stream does not take) compiled for host fibers (kernel_prelude.h); takes the place of
// convsep_f32.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#include "../../libvips_amd/csrc/convsep_f32.hip"
