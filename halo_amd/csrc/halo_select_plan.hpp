// Host-side interface between the two selector translation units: halo_select.hip owns the C entry point
// and the serial kernel, halo_select_binned.hip the value-binned sweep.
#pragma once
#include "halo_select_common.hpp"

namespace halo {

struct BinGeom {
    int H, W, n_regions, arad, mrad;
    unsigned kneed, captot, target, nfmax;
    int cs, gcy, gcx, gstride;   // pick grid: cell size mrad+1, padded row stride in bytes
    unsigned cmul;               // ceil(2^32 / cs): x / cs == mulhi(x, cmul) for x < 65536
    unsigned grid_bytes;
};

// Plan of one call: geometry, capacities and the workspace carve-up (offsets in bytes).
struct BinPlan {
    bool ok;                     // false: the sweep does not serve this geometry (serial kernel only)
    BinGeom g;
    size_t zero_bytes, total_bytes, lds_bytes;
    size_t off_hdr, off_hist1, off_fcur, off_okmin, off_okmax, off_cbase, off_cm, off_ckey, off_cpos, off_plist;
};

BinPlan binned_plan(int64_t B, int64_t H, int64_t W, int64_t n_regions, int64_t arad, int64_t mrad);

// Enqueue the binned selector.  Images it could not finish are marked SEL_BAIL in their SelHdr
// (`*hdr_out`, B entries) for the serial kernel to continue.
int binned_select(void *score, int dtype, int64_t B, const BinPlan &p, uint8_t *active, uint8_t *selected, int64_t *active_mask,
                  const int64_t *gt, double *picks, int32_t *n_picked, void *workspace, size_t workspace_bytes, hipStream_t st,
                  SelHdr **hdr_out, const void *score_range = nullptr, int32_t *handover = nullptr);

// Exact value range of B score maps (hw pixels each) as range records (one SelHdr-sized record per image).
int score_range_exact(const void *score, int dtype, int64_t B, int64_t hw, void *range_out, hipStream_t st);

}  // namespace halo
