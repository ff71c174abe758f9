// TEST INFRASTRUCTURE: libvips_amd/csrc/colour.hip ITSELF -- every colour route, the LDS-table sRGB -> Lab
// kernel, cast, premultiply / unpremultiply, vips_sharpen in one kernel -- compiled for host fibers
// (kernel_prelude.h); takes the place of colour.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#include "../../libvips_amd/csrc/colour.hip"
