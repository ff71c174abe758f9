// TEMPORARY: entry points declared in vips_hip.h whose kernels have not landed yet.
// Each fails loudly; none silently falls back to anything.
#include "internal.h"
using namespace vh;
#define TODO(name) do { error(name, "not implemented yet"); return -1; } while (0)
extern "C" {
VipsHipConv *vips_hip_conv_new(const double *, int, int, double, double, int) { error("conv", "not implemented yet"); return nullptr; }
void vips_hip_conv_free(VipsHipConv *) {}
int vips_hip_conv_get_nnz(const VipsHipConv *) { return -1; }
int vips_hip_conv_out_format(const VipsHipConv *, int) { return -1; }
int vips_hip_conv_gen(const VipsHipConv *, const VipsHipRegion *, const VipsHipRegion *) { TODO("conv"); }
int vips_hip_gaussmat(double, double, int, int, double *, int, double *) { TODO("gaussmat"); }
int vips_hip_colour_gen(int, const VipsHipRegion *, const VipsHipRegion *) { TODO("colour"); }
int vips_hip_cast_gen(const VipsHipRegion *, const VipsHipRegion *) { TODO("cast"); }
int vips_hip_sharpen_gen(const int *, const VipsHipRegion *, const VipsHipRegion *, const VipsHipRegion *) { TODO("sharpen"); }
int vips_hip_conv(VipsHipImage *, VipsHipImage **, const double *, int, int, double, double, int) { TODO("conv"); }
int vips_hip_convsep(VipsHipImage *, VipsHipImage **, const double *, int, double, double, int) { TODO("convsep"); }
int vips_hip_gaussblur(VipsHipImage *, VipsHipImage **, double, double, int) { TODO("gaussblur"); }
int vips_hip_sharpen(VipsHipImage *, VipsHipImage **, double, double, double, double, double, double) { TODO("sharpen"); }
int vips_hip_colourspace(VipsHipImage *, VipsHipImage **, int) { TODO("colourspace"); }
int vips_hip_cast(VipsHipImage *, VipsHipImage **, int) { TODO("cast"); }
}
