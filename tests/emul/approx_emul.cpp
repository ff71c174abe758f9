// TEST INFRASTRUCTURE: libvips_amd/csrc/approx.hip ITSELF compiled for host fibers (kernel_prelude.h); takes the
// place of approx.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#include "../../libvips_amd/csrc/approx.hip"
