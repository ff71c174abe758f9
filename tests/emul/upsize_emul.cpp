// TEST INFRASTRUCTURE: libvips_amd/csrc/upsize.hip ITSELF compiled for host fibers (kernel_prelude.h); takes the
// place of upsize.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#include "../../libvips_amd/csrc/upsize.hip"
