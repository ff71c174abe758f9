// TEST INFRASTRUCTURE: libvips_amd/csrc/conv.hip ITSELF compiled for host fibers (kernel_prelude.h); takes the
// place of conv.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#include "../../libvips_amd/csrc/conv.hip"
