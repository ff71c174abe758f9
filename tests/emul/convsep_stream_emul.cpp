// TEST INFRASTRUCTURE: libvips_amd/csrc/convsep_stream.hip ITSELF -- both passes of a separable float
// convolution, the colour epilogue, the integer horizontal pass: BASELINE config 3's kernel -- compiled
// for host fibers (kernel_prelude.h); takes the place of convsep_stream.hip in libvipship_emul.so.
#include "kernel_prelude.h"

#include "../../libvips_amd/csrc/convsep_stream.hip"
