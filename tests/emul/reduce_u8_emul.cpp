// TEST INFRASTRUCTURE: libvips_amd/csrc/reduce_u8.hip ITSELF -- the fused vips_reduce of BASELINE config 2 on the
// matrix instruction (v_mfma_f32_4x4x4_16b_f16 and the quad DPP moves as wave meetings of the fibers), the VALU
// sibling, the one-axis uchar kernels -- compiled for host fibers (kernel_prelude.h); takes the place of
// reduce_u8.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#define dot2 reduce_u8_dot2 // (the file has its own; this directory's gcn.h has one of the same name)

#include "../../libvips_amd/csrc/reduce_u8.hip"
