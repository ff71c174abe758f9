// TEST INFRASTRUCTURE: libvips_amd/csrc/resample_f32.hip ITSELF compiled for host fibers (kernel_prelude.h); takes
// the place of resample_f32.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#include "../../libvips_amd/csrc/resample_f32.hip"
