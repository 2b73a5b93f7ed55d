// Internal structures shared by the ORB kernels and the C-ABI implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/orbhip.h"

#define ORB_MAX_LEVELS 16
#define ORB_EDGE 19           // EDGE_THRESHOLD  (ORBextractor.cc:72)
#define ORB_MINB 16           // minBorder = EDGE_THRESHOLD-3 (ORBextractor.cc:771)
#define ORB_HALF_PATCH 15

// One pyramid level, batched: frame f lives at img + f*frame_stride, rows at pitch.
struct OrbLevel {
    uint8_t *img;             // un-padded level image (level 0 may alias the caller's input)
    uint8_t *blur;            // 7x7 sigma-2 blurred level
    size_t img_frame_stride;
    size_t blur_frame_stride;
    int w, h, img_pitch, blur_pitch;
    // FAST cell grid (ORBextractor.cc:774-781)
    int ncols, nrows, wcell, hcell;
    int cell_base;            // first cell of this level in the per-frame cell array
    int cell_cap;             // entries per cell list
    // octree
    int quota;                // mnFeaturesPerLevel[level]
    int key_base, key_cap;    // ordered-candidate scratch slice (per frame)
    int kp_base, kp_cap;      // per-level keypoint staging slice (per frame)
    int n_ini;                // nIni (ORBextractor.cc:541)
    float hx;                 // hX   (ORBextractor.cc:543)
    float scale;              // mvScaleFactor[level]
    float size;               // (int)(PATCH_SIZE*scale) as float (ORBextractor.cc:862)
    // resize tables (device): xofs[w], xalpha[2w], yofs[h], ybeta[2h]  (level>0)
    const int16_t *xofs; const int16_t *xalpha; const int16_t *yofs; const int16_t *ybeta;
    // k_resize_rows tables (level>0; xchunk == nullptr: the source span of 4 output columns does not fit 8 bytes, k_resize is used):
    // xchunk[12 * c] = {load offset, sel[4], alpha[4], v_perm selector of the byte shift, 0, 0} per 4 output columns, ytab[dy] = {row0, row1, b0 << 12, b1 << 12}
    const uint32_t *xchunk; const uint32_t *ytab;
    int resize_mode;          // k_resize_rows<MODE>: 0 unaligned 8-byte loads, 1 aligned dwordx3 + byte shift
};

struct OrbParams {
    OrbLevel lv[ORB_MAX_LEVELS];
    int nlevels, batch;
    int n_cus;                // compute units of the context's device
    int ini_th, min_th;
    int cells_per_frame;      // sum over levels of ncols*nrows
    int keys_per_frame;       // sum of key_cap
    int kps_per_frame;        // sum of kp_cap  (== staging slots per frame)
    int max_kp;               // output row capacity per frame
    int oct_nc;               // k_octree: node capacity of its LDS arrays = max over levels of max(quota + 16, 4*nIni + 4)
    int rows_min_batch;       // the row-streaming pyramid / blur kernels are used from this batch size on (below it the tile kernels have the shorter latency)
    int bm_min_batch;         // the matrix-core blur (k_blur_mfma) is used from this batch size on (its tables exist when bm_cols[nlevels] > 0)
    int br_blocks[ORB_MAX_LEVELS + 1];  // k_blur_rows: prefix of 256-lane workgroups per frame over the levels; [nlevels] == 0: k_blur (tiles) is used
    int bs_tiles[ORB_MAX_LEVELS + 1];   // k_blur: prefix of 64x32 tiles per frame over the levels (one launch for all levels)
    int lap0, lap1;
    // per-frame scratch
    uint32_t *cell_count;     // [batch][cells_per_frame]
    uint32_t *cell_list;      // [batch][sum(cells*cell_cap)]  packed x | y<<12 | score<<24
    size_t cell_list_frame_stride;
    uint32_t *keys;           // [batch][keys_per_frame] ordered candidates (packed)
    uint16_t *node_of;        // [batch][keys_per_frame]
    uint32_t *lvl_kp;         // [batch][kps_per_frame] selected keypoints (packed)
#define ORB_PERM_MIN_BATCH 16   // below this batch size k_orient_desc keeps the slot order (single-frame latency)
    uint16_t *lvl_perm;       // [batch][kps_per_frame] per level: position (inside the level slice) of the r-th keypoint in spatial order -- the order
                              // k_orient_desc WORKS in (patches of neighbours share cache lines); results stay in the reference's slots
    float *lvl_angle;         // [batch][kps_per_frame]
    uint8_t *lvl_desc;        // [batch][kps_per_frame][32]
    int32_t *lvl_count;       // [batch][nlevels]   post-octree count
    int32_t *lvl_ncand;       // [batch][nlevels]   pre-octree count
    int32_t *status;          // [1] sticky error flag (capacity overflow)
    // outputs
    orbhip_keypoint *out_kp;  // [batch][max_kp]
    uint8_t *out_desc;        // [batch][max_kp][32]
    int32_t *out_count;       // [batch]
    int32_t *out_mono;        // [batch]
    int umax[ORB_HALF_PATCH + 1];
    int gauss_q8[7];
    // k_blur_mfma (round 3): the 7x7 blur as two banded int8 matrix products per 64 x 64 window, one wave per 32-column tile column
    const uint4 *bm_th;                 // [tile columns of all levels][2 output blocks][64 lanes] horizontal band operands, image borders folded in
    const uint4 *bm_tv;                 // [4 output blocks][64 lanes] vertical band operands
    int bm_cols[ORB_MAX_LEVELS + 1];    // prefix of tile columns per frame over the levels; [nlevels] == 0: the kernel is not used
    int bm_init;                        // 128 * (sum of the taps): turns the biased int8 sums back into the plain ones
};

// k_fast_cells has its own, compact parameter block (scalar loads of per-level fields are what a persistent wave does most)
struct FcLevel {
    const uint8_t *img; size_t frame_stride; int img_pitch;
    int max_bx, max_by;           // w - minBorder, h - minBorder
    int ncols, nrows, wcell, hcell, cell_base, cell_cap;
    int tpr, rpt;                 // lane layout of the per-pair passes: two-pair tasks per row (nominal cell width), rows per trip
    int rpc, rpr, run_base;       // k_fast_runs: cells per run (2 while two cells fit 64 pixels, else 1), runs per cell row, first run of this level in a frame's run array
};
struct FastParams {
    FcLevel lv[ORB_MAX_LEVELS];
    int nlevels, batch, ini_th, min_th, cells_per_frame;
    int n_cus;                    // compute units of the context's device (hipDeviceAttributeMultiprocessorCount): sizes the persistent grid
    int chunk;                    // consecutive cells a wave takes at a time (set per launch)
    int lvl_lo, lvl_hi;           // this launch covers the cells of levels [lvl_lo, lvl_hi) of every frame (level 0 can start before the pyramid exists)
    uint32_t *cell_count, *cell_list; size_t cell_list_frame_stride; int32_t *status;
    // per-wave LDS geometry (from the largest cell of this extractor): pair tile [rows][PITCH] dwords, score tile [srows + 2][SPITCH]
    // dwords, queue of qcap u16; wave_dw = dwords per wave.  small_cells: every level fits the <28, 24, 9> instantiation
    int rows, srows, qcap, wave_dw, small_cells;
    uint32_t div_magic[34];       // floor(i / n) == (i * div_magic[n]) >> 16 whenever i * n < 65536, n = 1 .. 33
    // k_fast_runs (round 4): runs of up to two horizontally adjacent cells per wave
    int runs_per_frame, run_rows, run_q0, run_dw, use_runs;      // LDS rows of the pair tile, entries of the first queue, dwords per wave
};

#define ORB_PACK_KEY(x, y, s) ((uint32_t)(x) | ((uint32_t)(y) << 12) | ((uint32_t)(s) << 24))
#define ORB_KEY_X(k) ((int)((k) & 0xFFFu))
#define ORB_KEY_Y(k) ((int)(((k) >> 12) & 0xFFFu))
#define ORB_KEY_S(k) ((int)((k) >> 24))

// Frame::ComputeStereoMatches (stereo_kernels.hip): the two extractors' pyramids and device outputs
struct StereoLevel {
    const uint8_t *imgL, *imgR;
    size_t fsL, fsR;              // frame strides
    int pitchL, pitchR, wR, pad_;
    float scale, inv_scale;       // mvScaleFactors / mvInvScaleFactors of the LEFT extractor
};
struct StereoArgs {
    StereoLevel lv[ORB_MAX_LEVELS];
    int nlevels, batch, max_kp, rows0;
    const orbhip_keypoint *kpL, *kpR;
    const uint8_t *descL, *descR;
    const int32_t *nL, *nR;
    float mb, mbf;
    float *u_right, *depth;
    int32_t *sad, *n_kept;
};
void orb_launch_stereo(const StereoArgs &A, hipStream_t s);

// Opt a kernel into the largest dynamic-LDS size a workgroup may have (160 KB minus its static LDS), once per (kernel, device);
// thread-safe, never lowers the limit again.  `need` = the dynamic bytes of the coming launch: more than the limit -> error.
int orb_lds_optin(const void *func, int device, size_t need);

// kernel launchers (orb_kernels.hip)
void orb_launch_resize(const OrbParams &P, int level, hipStream_t s);
void orb_launch_fast_cells(const FastParams &F, hipStream_t s, int max_per_cu, int lvl_lo = 0, int lvl_hi = -1);      // max_per_cu: 0 = as many waves as fit
const void *orb_fast_runs_func(int nld);
#define ORB_OVERLAP_MIN_BATCH 16
#ifndef BLR_R
#define BLR_R 24              // k_blur_rows: output rows per lane (multiple of 6)
#endif
void orb_launch_blur(const OrbParams &P, hipStream_t s, int wgs_per_cu, int frame0 = 0, int nframes = -1);   // row kernel: frames [frame0, frame0 + nframes)
const void *orb_fast_cells_func(int small);
void orb_launch_octree(const OrbParams &P, hipStream_t s);
void orb_launch_orient_desc(const OrbParams &P, hipStream_t s);
void orb_launch_assemble(const OrbParams &P, hipStream_t s);
size_t orb_octree_lds_bytes(int nc);
int orb_octree_nc(const OrbParams &P);
const void *orb_octree_func();
