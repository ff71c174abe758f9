// TEST INFRASTRUCTURE: libvips_amd/csrc/resize_stream.hip ITSELF (the vips_resize chains of BASELINE configs 1 and 4 in
// one kernel) compiled for host fibers (kernel_prelude.h); takes the place of resize_stream.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#include "../../libvips_amd/csrc/resize_stream.hip"
