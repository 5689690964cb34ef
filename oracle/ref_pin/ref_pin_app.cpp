// oracle/_ref/libref_pin.so, second translation unit: the APPLICATION's own constants, read from the reference's 360_stitcher/defs.h compiled where it lies
// (the header is self-contained: <string>, <vector>, <iostream>, <chrono> and plain const definitions).  Pins -- by compiling the reference's own file -- the values the
// drop-in surface restates: the shipped configuration of bench.py / stitch_app (WORK / SEAM / COMPOSE megapixels, NUM_IMAGES, OUTPUT size, BLEND_STRENGTH, mesh size),
// the shim's recalibration constants (RECALIB_THRESH, MAX_FEATURES_PER_IMAGE) and the mesh optimiser's defaults (ALPHAS, GLOBAL_DIST).  Test infrastructure only.
#include <cstring>
#include "defs.h"

extern "C" {

// integer constants by name; returns 0 and leaves *out untouched for an unknown name
int ref_pin_app_int(const char *name, long long *out)
{
#define K(n) if (!std::strcmp(name, #n)) { *out = (long long)(n); return 1; }
    K(NUM_IMAGES) K(OUTPUT_WIDTH) K(OUTPUT_HEIGHT) K(RECALIB_DEL) K(RECALIB_THRESH) K(RECALIB_INTERP) K(HESS_THRESH) K(NOCTAVES) K(NOCTAVESLAYERS)
    K(MAX_FEATURES_PER_IMAGE) K(MAX_TEMPORAL_FEATURES_PER_IMAGE) K(MESH_HEIGHT) K(MESH_WIDTH) K(GLOBAL_DIST) K(CAPTURE_IMG_WIDTH) K(CAPTURE_IMG_HEIGHT)
    K(CAPTURE_IMG_CHANNELS) K(wrapAround) K(recalibrate) K(enable_local) K(keep_aspect_ratio) K(add_black_bars) K(USE_TEMPORAL) K(skip_frames)
#undef K
    return 0;
}
int ref_pin_app_double(const char *name, double *out)
{
#define K(n) if (!std::strcmp(name, #n)) { *out = (double)(n); return 1; }
    K(WORK_MEGAPIX) K(SEAM_MEAGPIX) K(COMPOSE_MEGAPIX) K(MATCH_CONF) K(BLEND_STRENGTH) K(PI)
#undef K
    if (!std::strncmp(name, "ALPHAS", 6) && name[6] >= '0' && name[6] <= '3' && !name[7]) { *out = (double)ALPHAS[name[6] - '0']; return 1; }
    return 0;
}
int ref_pin_app_offset(int i) { return (i >= 0 && i < NUM_IMAGES) ? offsets[i] : -1; }

}  // extern "C"
