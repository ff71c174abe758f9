// Internal: state of one reduceh/reducev operation.
#pragma once

#include "internal.h"

#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace vh {

// include/vips/interpolate.h:109-118, resample/presample.h:70
constexpr int TRANSFORM_SHIFT = 6;
constexpr int TRANSFORM_SCALE = 1 << TRANSFORM_SHIFT;
constexpr int INTERPOLATE_SHIFT = 12;
constexpr int INTERPOLATE_SCALE = 1 << INTERPOLATE_SHIFT;
constexpr int MAX_POINT = 2000;

// Where output sample k reads from: first tap in UN-embedded input coordinates
// (may be negative / beyond the edge: taps are clamped) and coefficient phase.
struct ReducePos {
	int first;
	int phase;
};

// one coefficient of a long double mask (x87 extended: 64 bits of mantissa) as the device's
// soft-float takes it: value = (-1)^sign * mant * 2^(exp - 63), mant normalised (bit 63) or 0
struct ReduceTap80 {
	unsigned long long mant;
	int exp;
	int sign;
};

} // namespace vh

struct _VipsHipReduce {
	int kernel;
	double shrink; // residual shrink
	int in_size, out_size;
	int n_point;
	double offset; // hoffset / voffset
	int embed;     // ceil(n_point / 2) - 1
	std::vector<double> matrixf;
	std::vector<short> matrixs;
	double *d_matrixf;
	short *d_matrixs;
	// device copies of position arrays keyed by (start, count, tile)
	std::map<std::tuple<int, int, int>, vh::ReducePos *> pos_cache;
	// what the host keeps about a pos_cache blob (a few integers: the streaming kernels' geometry), by blob;
	// lives and dies with the plan (round 4 kept it in a function-static map that outlived freed plans)
	std::map<const void *, std::vector<long long>> blob_info;
	std::mutex mutex;
	mutable std::atomic<int> device{ -1 }; // where the device tables live (vh::plan_device)
};

namespace vh {

void reduce_make_mask(double *c, int kernel, int n_points, double shrink, double x);
// float images as streams (resample_f32.hip): 1 launched, 0 not their case, -1 error
int reducev_f32_stream_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const std::vector<ReducePos> &pos, const double *coef /* device: the plan's double table */);
int reduceh_f32_lds_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const std::vector<ReducePos> &pos, const double *coef);
int shrinkv_f32_stream_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out);
void reduce_positions(const _VipsHipReduce *r, int start, int count, int tile,
	std::vector<ReducePos> &pos);
// positions and per-output long double masks of the double-image path (resample_host.cpp)
void reduce_notab_masks(const _VipsHipReduce *r, int start, int count, int tile, std::vector<ReducePos> &pos,
	std::vector<ReduceTap80> &taps);

} // namespace vh
