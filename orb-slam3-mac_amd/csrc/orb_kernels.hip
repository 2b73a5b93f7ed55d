// orb_kernels.hip -- hand-written gfx950 kernels for the ORB front-end
// (ORBextractor::operator(), reference src/ORBextractor.cc:1068-1150).
//
// One launch processes a whole batch of frames; blockIdx carries (tile|cell|slot, frame).
// Integer/byte work throughout: HBM/LDS/VALU-bound, no MFMA (nothing here is GEMM-shaped).
// Wave = 64 lanes everywhere.  Compile: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include "orb_internal.h"

#define WAVE 64

// ----------------------------------------------------------------------------------
// helpers
// ----------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of one int per thread over a 256-thread block. `wsum` = 8 ints of LDS.
// Returns exclusive prefix; *total = block sum.  Contains two __syncthreads().
__device__ __forceinline__ int block_excl_scan256(int v, int *wsum, int *total)
{
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    int inc = wave_incl_scan(v);
    if (lane == WAVE - 1) wsum[wid] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        int s = wsum[w];
        if (w < wid) base += s;
    }
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - v;
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

// ----------------------------------------------------------------------------------
// A2/A3  pyramid: level l from level l-1, cv::resize INTER_LINEAR 8UC1 fixed point
// (reference call ORBextractor.cc:1165; arithmetic SURVEY Appendix A.3).
// Each thread produces 4 horizontally adjacent output pixels and stores one dword.
// Coefficient tables are computed on the host in float exactly like OpenCV does.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(OrbParams P, int level)
{
    const OrbLevel &D = P.lv[level];
    const OrbLevel &S = P.lv[level - 1];
    const int frame = blockIdx.z;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy >= D.h || dx0 >= D.w) return;
    const int sy = D.yofs[dy];
    const int b0 = D.ybeta[2 * dy], b1 = D.ybeta[2 * dy + 1];
    const int y0 = sy < 0 ? 0 : (sy < S.h ? sy : S.h - 1);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 < S.h ? sy + 1 : S.h - 1);
    const uint8_t *src = S.img + (size_t)frame * S.img_frame_stride;
    const uint8_t *S0 = src + (size_t)y0 * S.img_pitch;
    const uint8_t *S1 = src + (size_t)y1 * S.img_pitch;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = dx0 + i;
        if (dx < D.w) {
            const int sx = D.xofs[dx];
            const int sx1 = sx + 1 < S.w ? sx + 1 : sx;
            const int a0 = D.xalpha[2 * dx], a1 = D.xalpha[2 * dx + 1];
            const int t0 = S0[sx] * a0 + S0[sx1] * a1;
            const int t1 = S1[sx] * a0 + S1[sx1] * a1;
            const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 255) << (8 * i);
        }
    }
    uint8_t *dst = D.img + (size_t)frame * D.img_frame_stride + (size_t)dy * D.img_pitch + dx0;
    *reinterpret_cast<uint32_t *>(dst) = packed;   // pitch is a multiple of 64: pad bytes are scratch
}

void orb_launch_resize(const OrbParams &P, int level, hipStream_t s)
{
    const OrbLevel &D = P.lv[level];
    dim3 grid((D.w + 255) / 256, (D.h + 3) / 4, P.batch);
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, P, level);
}

// ----------------------------------------------------------------------------------
// A3/A4  per-cell FAST-9/16 + score + 3x3 NMS + two-threshold retry
// (ORBextractor.cc:783-854 calling cv::FAST twice; OpenCV FAST_t<16>/cornerScore<16>).
//
// Identity used (tests/test_oracle_orb.py::test_fast_arc_score_identity):
//   S = max over the 16 contiguous 9-arcs of min(v-p) and of min(p-v);
//   corner at threshold t  <=>  S > t ;  cornerScore == S-1.
// So one pass computes a threshold-independent score map; both thresholds are decided
// from it.  One 256-thread workgroup per cell, the cell's (wCell+6)x(hCell+6) sub-image
// staged in LDS.  Keypoints are emitted in cv::FAST order (row-major) via a block scan.
// ----------------------------------------------------------------------------------
#define FAST_TP 64            // LDS tile pitch (bytes); cells up to 64x64 incl. the 6-px apron
#define FAST_MAX_PPT 16       // pixels per thread upper bound: 58*58/256 < 16

__device__ __forceinline__ int fast_arc_score(const uint8_t *c)
{
    // Bresenham circle r=3 in cv::makeOffsets order.
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * FAST_TP + 0];   d[1] = v - c[3 * FAST_TP + 1];
    d[2] = v - c[2 * FAST_TP + 2];   d[3] = v - c[1 * FAST_TP + 3];
    d[4] = v - c[3];                 d[5] = v - c[-1 * FAST_TP + 3];
    d[6] = v - c[-2 * FAST_TP + 2];  d[7] = v - c[-3 * FAST_TP + 1];
    d[8] = v - c[-3 * FAST_TP + 0];  d[9] = v - c[-3 * FAST_TP - 1];
    d[10] = v - c[-2 * FAST_TP - 2]; d[11] = v - c[-1 * FAST_TP - 3];
    d[12] = v - c[-3];               d[13] = v - c[1 * FAST_TP - 3];
    d[14] = v - c[2 * FAST_TP - 2];  d[15] = v - c[3 * FAST_TP - 1];
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { lo2[i] = min(d[i], d[(i + 1) & 15]); hi2[i] = max(d[i], d[(i + 1) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; i++) { lo4[i] = min(lo2[i], lo2[(i + 2) & 15]); hi4[i] = max(hi2[i], hi2[(i + 2) & 15]); }
    int A = -256, B = 256;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int lo9 = min(min(lo4[i], lo4[(i + 4) & 15]), d[(i + 8) & 15]);
        int hi9 = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
        A = max(A, lo9);
        B = min(B, hi9);
    }
    return max(A, -B);
}

__global__ __launch_bounds__(256) void k_fast_cells(OrbParams P)
{
    __shared__ uint8_t tile[64 * FAST_TP];
    __shared__ uint8_t sc[66 * FAST_TP];      // score map, +1 row apron top/bottom
    __shared__ int wsum[8];
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int frame = blockIdx.y;
    const int cell = blockIdx.x;
    int lvl = 0;
    for (int l = 1; l < P.nlevels; l++) if (cell >= P.lv[l].cell_base) lvl = l;
    const OrbLevel &L = P.lv[lvl];
    const int c = cell - L.cell_base;
    const int ci = c / L.ncols, cj = c - ci * L.ncols;
    uint32_t *cnt_out = P.cell_count + (size_t)frame * P.cells_per_frame + cell;
    const int max_bx = L.w - ORB_MINB, max_by = L.h - ORB_MINB;
    const int ini_y = ORB_MINB + ci * L.hcell, ini_x = ORB_MINB + cj * L.wcell;
    int max_y = ini_y + L.hcell + 6, max_x = ini_x + L.wcell + 6;
    if (ini_y >= max_by - 3 || ini_x >= max_bx - 6) {       // ORBextractor.cc:788,797
        if (tid == 0) *cnt_out = 0;
        return;
    }
    if (max_y > max_by) max_y = max_by;
    if (max_x > max_bx) max_x = max_bx;
    const int cw = max_x - ini_x, ch = max_y - ini_y;
    const int dw = cw - 6, dh = ch - 6;                       // detection band of cv::FAST
    if (dw <= 0 || dh <= 0) {
        if (tid == 0) *cnt_out = 0;
        return;
    }
    // stage sub-image, clear score map
    const uint8_t *img = L.img + (size_t)frame * L.img_frame_stride + (size_t)ini_y * L.img_pitch + ini_x;
    for (int i = tid; i < cw * ch; i += 256) {
        const int y = i / cw, x = i - y * cw;
        tile[y * FAST_TP + x] = img[(size_t)y * L.img_pitch + x];
    }
    for (int i = tid; i < 66 * FAST_TP / 4; i += 256) reinterpret_cast<uint32_t *>(sc)[i] = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // score map over the detection band; sc index (y+1, x) so the apron rows exist
    const int npix = dw * dh;
    for (int i = tid; i < npix; i += 256) {
        const int y = 3 + i / dw, x = 3 + i % dw;
        const int S = fast_arc_score(&tile[y * FAST_TP + x]);
        sc[(y + 1) * FAST_TP + x] = (uint8_t)(S > P.min_th ? S - 1 : 0);
    }
    __syncthreads();
    // NMS, thread owns a contiguous run of pixels (row-major) -> ordered emission
    const int ppt = (npix + 255) / 256;
    const int p0 = tid * ppt;
    uint32_t keep = 0;
    int th = P.ini_th;
    for (int pass = 0; pass < 2; pass++) {
        keep = 0;
        for (int k = 0; k < ppt; k++) {
            const int i = p0 + k;
            if (i < npix) {
                const int y = 3 + i / dw, x = 3 + i % dw;
                const uint8_t *q = &sc[(y + 1) * FAST_TP + x];
                // v(th) = score if score >= th (i.e. S > th) else 0
                const int v = q[0] >= th ? q[0] : 0;
                if (v) {
                    int m = 0;
#pragma unroll
                    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                        for (int dx = -1; dx <= 1; dx++)
                            if (dx | dy) { const int nb = q[dy * FAST_TP + dx]; m = max(m, nb >= th ? nb : 0); }
                    if (v > m) keep |= 1u << k;
                }
            }
        }
        if (pass == 0) {
            if (keep) atomicAdd(&s_cnt, 1);
            __syncthreads();
            if (s_cnt > 0) break;                          // vKeysCell non-empty at iniThFAST
            th = P.min_th;                                 // ORBextractor.cc:825-828 retry
        }
    }
    int total;
    int off = block_excl_scan256(__popc(keep), wsum, &total);
    uint32_t *list = P.cell_list + (size_t)frame * P.cell_list_frame_stride + (size_t)cell * L.cell_cap;
    if (total > L.cell_cap) { if (tid == 0) atomicExch(P.status, ORBHIP_E_CAPACITY); total = L.cell_cap; }
    for (int k = 0; k < ppt; k++) {
        if (keep & (1u << k)) {
            const int i = p0 + k;
            const int y = 3 + i / dw, x = 3 + i % dw;
            if (off < L.cell_cap)
                list[off] = ORB_PACK_KEY(x + cj * L.wcell, y + ci * L.hcell, sc[(y + 1) * FAST_TP + x]);
            off++;
        }
    }
    if (tid == 0) *cnt_out = (uint32_t)total;
}

void orb_launch_fast(const OrbParams &P, hipStream_t s)
{
    dim3 grid(P.cells_per_frame, P.batch);
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, P);
}

// ----------------------------------------------------------------------------------
// A5  DistributeOctTree (ORBextractor.cc:537-761), one 256-thread workgroup per
// (frame, level), node state resident in LDS.
//
// Reformulation (bit-identical to the oracle's list-based restatement):
//  * every node's vKeys is a subsequence of the level's candidate list in original
//    order (DivideNode pushes in order), so a per-key node id replaces the key vectors;
//  * the std::list order is an explicit position: a step that splits the set `proc`
//    in processing order o=0..m-1 yields  [children(o=m-1) n4..n1, ..., children(o=0)
//    n4..n1, surviving old nodes in old order]  (push_front semantics);
//  * full pass: proc = all nodes with >1 key, processing order = list order;
//  * final phase (ORBextractor.cc:665-735): candidates = children with >1 key created by
//    the previous step, processing order = (size desc, creation desc) == (size desc,
//    list position asc); stop after the first split that reaches N nodes.  The
//    reference breaks ties by node address (non-deterministic); oracle and
//    kernel use creation order -- documented deviation.
// ----------------------------------------------------------------------------------
struct OctLds {
    uint32_t *cc;        // [NC*4] child key counts
    int *ord;            // [NC] processing order of node p, or -1
    int *by_ord;         // [NC] node at order o
    int *excl;           // [NC] exclusive scan over order of nchild
    int *posbase;        // [NC] new position of the first (frontmost) child of processed p
    int *newpos;         // [NC] new position of a surviving node p
    uint32_t *best;      // [NC]
};

size_t orb_octree_lds_bytes(int max_quota)
{
    int nc = max_quota + 16;
    return (size_t)nc * 4 * (6 + 4 + 6) + 64;
}

__device__ __forceinline__ int oct_quadrant(uint32_t key, uint32_t b0, uint32_t b1)
{
    // DivideNode (ORBextractor.cc:479-524): halfX = ceil((UR.x-UL.x)/2), children by < on x then y
    const int x0 = b0 & 0xFFFF, y0 = b0 >> 16, x1 = b1 & 0xFFFF, y1 = b1 >> 16;
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
    const int kx = ORB_KEY_X(key), ky = ORB_KEY_Y(key);
    return (kx < mx ? 0 : 1) + (ky < my ? 0 : 2);          // n1=0 n2=1 n3=2 n4=3
}

__global__ __launch_bounds__(256) void k_octree(OrbParams P)
{
    extern __shared__ uint32_t smem[];
    __shared__ int wsum[8];
    __shared__ int s_size, s_front, s_nexpand, s_rstar, s_T, s_nproc;
    const int tid = threadIdx.x;
    const int lvl = blockIdx.x / P.batch;              // level-major: big levels first
    const int frame = blockIdx.x - lvl * P.batch;
    const OrbLevel &L = P.lv[lvl];
    const int N = L.quota;
    const int NC = [&] { int m = 0; for (int l = 0; l < P.nlevels; l++) m = max(m, P.lv[l].quota); return m + 16; }();
    // double-buffered node arrays (by list position): box0 = UL.x|UL.y<<16, box1 = BR.x|BR.y<<16, cnt = #keys
    uint32_t *box0 = smem, *box1 = smem + NC, *cnt = smem + 2 * NC;
    uint32_t *nbox0 = smem + 3 * NC, *nbox1 = smem + 4 * NC, *ncnt = smem + 5 * NC;
    OctLds S;
    {
        uint32_t *p = smem + 6 * NC;
        S.cc = p; p += 4 * NC;
        S.ord = (int *)p; p += NC; S.by_ord = (int *)p; p += NC; S.excl = (int *)p; p += NC;
        S.posbase = (int *)p; p += NC; S.newpos = (int *)p; p += NC; S.best = p; p += NC;
    }
    // ---- gather candidates in reference order: cells row-major, list order inside a cell
    const uint32_t *ccount = P.cell_count + (size_t)frame * P.cells_per_frame + L.cell_base;
    const uint32_t *clist = P.cell_list + (size_t)frame * P.cell_list_frame_stride + (size_t)L.cell_base * L.cell_cap;
    uint32_t *keys = P.keys + (size_t)frame * P.keys_per_frame + L.key_base;
    uint16_t *node_of = P.node_of + (size_t)frame * P.keys_per_frame + L.key_base;
    const int ncells = L.ncols * L.nrows;
    int running = 0;
    for (int c0 = 0; c0 < ncells; c0 += 256) {
        const int c = c0 + tid;
        const int n = c < ncells ? (int)ccount[c] : 0;
        int total;
        const int off = running + block_excl_scan256(n, wsum, &total);
        for (int i = 0; i < n; i++)
            if (off + i < L.key_cap) keys[off + i] = clist[(size_t)c * L.cell_cap + i];
        running += total;
    }
    if (running > L.key_cap) { if (tid == 0) atomicExch(P.status, ORBHIP_E_CAPACITY); running = L.key_cap; }
    const int K = running;
    if (tid == 0) P.lvl_ncand[frame * P.nlevels + lvl] = K;
    int32_t *count_out = P.lvl_count + frame * P.nlevels + lvl;
    if (K == 0) { if (tid == 0) *count_out = 0; return; }
    __syncthreads();                                     // keys[] visible block-wide (same CU, L1 coherent within WG)
    __threadfence_block();

    // ---- roots (ORBextractor.cc:541-584)
    const int n_ini = L.n_ini;
    const int H = (L.h - ORB_MINB) - ORB_MINB;
    for (int i = tid; i < NC; i += 256) { cnt[i] = 0; ncnt[i] = 0; }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        const int r = (int)__fdiv_rn((float)ORB_KEY_X(keys[k]), L.hx);   // ORBextractor.cc:568
        node_of[k] = (uint16_t)r;
        atomicAdd(&ncnt[r], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        // compact non-empty roots, keep order; remap handled below through newpos
        int m = 0;
        for (int i = 0; i < n_ini; i++) {
            const uint32_t c = ncnt[i];
            S.newpos[i] = m;
            if (c) {
                box0[m] = (uint32_t)(int)__fmul_rn(L.hx, (float)i);              // UL.x | 0<<16
                box1[m] = (uint32_t)(int)__fmul_rn(L.hx, (float)(i + 1)) | ((uint32_t)H << 16);
                cnt[m] = c;
                m++;
            }
        }
        s_size = m; s_front = m;
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) node_of[k] = (uint16_t)S.newpos[node_of[k]];
    __syncthreads();

    // ---- subdivision loop
    bool final_phase = false;
    for (int guard = 0; guard < 64; guard++) {
        const int size = s_size, front = s_front;
        // 1. candidate set + processing order
        int C;   // number of candidates
        if (!final_phase) {
            int run = 0;
            for (int p0 = 0; p0 < size; p0 += 256) {
                const int p = p0 + tid;
                const int f = (p < size && cnt[p] > 1) ? 1 : 0;
                int tot;
                const int o = run + block_excl_scan256(f, wsum, &tot);
                if (p < size) { S.ord[p] = f ? o : -1; if (f) S.by_ord[o] = p; }
                run += tot;
            }
            C = run;
        } else {
            // candidates: p < front with cnt > 1 ; order by (cnt desc, p asc)
            int run = 0;
            for (int p0 = 0; p0 < size; p0 += 256) {
                const int p = p0 + tid;
                const int f = (p < front && cnt[p] > 1) ? 1 : 0;
                int tot;
                const int o = run + block_excl_scan256(f, wsum, &tot);
                if (p < size) S.ord[p] = -1;
                if (f) S.excl[o] = p;                   // temp: compacted candidate list (ascending p)
                run += tot;
            }
            C = run;
            __syncthreads();
            for (int i = tid; i < C; i += 256) {
                const int p = S.excl[i];
                const uint32_t ci = cnt[p];
                int r = 0;
                for (int j = 0; j < C; j++) {
                    const uint32_t cj = cnt[S.excl[j]];
                    r += (cj > ci || (cj == ci && j < i)) ? 1 : 0;
                }
                S.ord[p] = r;
                S.by_ord[r] = p;
            }
        }
        __syncthreads();
        if (C == 0) break;                               // size == prevSize -> bFinish
        // 2. child key counts of every candidate
        for (int i = tid; i < 4 * size; i += 256) S.cc[i] = 0;
        if (tid == 0) { s_nexpand = 0; s_rstar = C - 1; }
        __syncthreads();
        for (int k = tid; k < K; k += 256) {
            const int p = node_of[k];
            if (S.ord[p] >= 0) atomicAdd(&S.cc[4 * p + oct_quadrant(keys[k], box0[p], box1[p])], 1u);
        }
        __syncthreads();
        // 3. exclusive scan over order of the non-empty-children counts
        {
            int run = 0;
            for (int o0 = 0; o0 < C; o0 += 256) {
                const int o = o0 + tid;
                int nch = 0;
                if (o < C) { const uint32_t *q = &S.cc[4 * S.by_ord[o]]; nch = (q[0] > 0) + (q[1] > 0) + (q[2] > 0) + (q[3] > 0); }
                int tot;
                const int e = run + block_excl_scan256(nch, wsum, &tot);
                if (o < C) {
                    S.excl[o] = e;
                    // final phase: stop after the first split that reaches N (ORBextractor.cc:727-728)
                    if (final_phase && size + e + nch - (o + 1) >= N) atomicMin(&s_rstar, o);
                }
                run += tot;
            }
        }
        __syncthreads();
        const int rstar = s_rstar;
        if (tid == 0) {
            const uint32_t *q = &S.cc[4 * S.by_ord[rstar]];
            s_T = S.excl[rstar] + (q[0] > 0) + (q[1] > 0) + (q[2] > 0) + (q[3] > 0);
            s_nproc = rstar + 1;
        }
        __syncthreads();
        const int T = s_T, nproc = s_nproc;
        // 4. positions of surviving nodes: T + rank among non-processed (old order)
        {
            int run = 0;
            for (int p0 = 0; p0 < size; p0 += 256) {
                const int p = p0 + tid;
                const int surv = (p < size && !(S.ord[p] >= 0 && S.ord[p] <= rstar)) ? 1 : 0;
                int tot;
                const int e = run + block_excl_scan256(surv, wsum, &tot);
                if (p < size && surv) S.newpos[p] = T + e;
                run += tot;
            }
        }
        // 5. build the new node arrays
        for (int p = tid; p < size; p += 256) {
            const int o = S.ord[p];
            if (o >= 0 && o <= rstar) {
                const uint32_t *q = &S.cc[4 * p];
                const int nch = (q[0] > 0) + (q[1] > 0) + (q[2] > 0) + (q[3] > 0);
                const int base = T - (S.excl[o] + nch);          // frontmost child (n4 side)
                S.posbase[p] = base;
                const uint32_t b0 = box0[p], b1 = box1[p];
                const int x0 = b0 & 0xFFFF, y0 = b0 >> 16, x1 = b1 & 0xFFFF, y1 = b1 >> 16;
                const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                int pos = base, nexp = 0;
                for (int qd = 3; qd >= 0; qd--) {
                    if (q[qd]) {
                        const int cx0 = (qd & 1) ? mx : x0, cx1 = (qd & 1) ? x1 : mx;
                        const int cy0 = (qd & 2) ? my : y0, cy1 = (qd & 2) ? y1 : my;
                        nbox0[pos] = (uint32_t)cx0 | ((uint32_t)cy0 << 16);
                        nbox1[pos] = (uint32_t)cx1 | ((uint32_t)cy1 << 16);
                        ncnt[pos] = q[qd];
                        nexp += q[qd] > 1;
                        pos++;
                    }
                }
                if (nexp) atomicAdd(&s_nexpand, nexp);
            } else {
                const int np = S.newpos[p];
                nbox0[np] = box0[p]; nbox1[np] = box1[p]; ncnt[np] = cnt[p];
            }
        }
        __syncthreads();
        // 6. re-label keys
        for (int k = tid; k < K; k += 256) {
            const int p = node_of[k];
            const int o = S.ord[p];
            int np;
            if (o >= 0 && o <= rstar) {
                const int qd = oct_quadrant(keys[k], box0[p], box1[p]);
                const uint32_t *q = &S.cc[4 * p];
                int r = 0;                                         // non-empty siblings in front (q' > qd)
                for (int j = 3; j > qd; j--) r += q[j] > 0;
                np = S.posbase[p] + r;
            } else np = S.newpos[p];
            node_of[k] = (uint16_t)np;
        }
        __syncthreads();
        const int new_size = T + (size - nproc);
        const int nexpand = s_nexpand;
        __syncthreads();
        if (tid == 0) { s_size = new_size; s_front = T; }
        { uint32_t *t; t = box0; box0 = nbox0; nbox0 = t; t = box1; box1 = nbox1; nbox1 = t; t = cnt; cnt = ncnt; ncnt = t; }
        __syncthreads();
        // 7. termination (ORBextractor.cc:661-735)
        if (new_size >= N || new_size == size) break;
        if (!final_phase && new_size + 3 * nexpand > N) final_phase = true;
    }
    __syncthreads();
    // ---- best key per node: max response, first in list order wins (ORBextractor.cc:739-758)
    const int size = s_size;
    for (int i = tid; i < size; i += 256) S.best[i] = 0;
    __syncthreads();
    for (int k = tid; k < K; k += 256)
        atomicMax(&S.best[node_of[k]], ((uint32_t)ORB_KEY_S(keys[k]) << 20) | (uint32_t)(0xFFFFF - k));
    __syncthreads();
    uint32_t *out = P.lvl_kp + (size_t)frame * P.kps_per_frame + L.kp_base;
    int nout = size;
    if (nout > L.kp_cap) { if (tid == 0) atomicExch(P.status, ORBHIP_E_CAPACITY); nout = L.kp_cap; }
    for (int i = tid; i < nout; i += 256) out[i] = keys[0xFFFFF - (S.best[i] & 0xFFFFF)];
    if (tid == 0) *count_out = nout;
}

void orb_launch_octree(const OrbParams &P, hipStream_t s)
{
    int mq = 0;
    for (int l = 0; l < P.nlevels; l++) mq = P.lv[l].quota > mq ? P.lv[l].quota : mq;
    size_t lds = orb_octree_lds_bytes(mq);
    hipLaunchKernelGGL(k_octree, dim3(P.nlevels * P.batch), dim3(256), lds, s, P);
}

// ----------------------------------------------------------------------------------
// A7  7x7 sigma-2 Gaussian, 8-bit fixed point, BORDER_REFLECT_101 (ORBextractor.cc:1114-1115;
// arithmetic SURVEY Appendix A.7).  Separable inside LDS: the row pass of an 8-bit image
// with the q8 kernel (sum 257) fits u16 exactly (255*257 = 65535), column pass
// (sum + 2^15) >> 16 saturated.  Tile: 64x32 outputs per 256-thread workgroup.
// ----------------------------------------------------------------------------------
#define BL_TW 64
#define BL_TH 32
__device__ __forceinline__ int reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void k_blur(OrbParams P, int level)
{
    __shared__ uint8_t in[(BL_TH + 6) * (BL_TW + 8)];
    __shared__ uint16_t hz[(BL_TH + 6) * BL_TW];
    const OrbLevel &L = P.lv[level];
    const int tid = threadIdx.x, frame = blockIdx.z;
    const int x0 = blockIdx.x * BL_TW, y0 = blockIdx.y * BL_TH;
    const uint8_t *src = L.img + (size_t)frame * L.img_frame_stride;
    for (int i = tid; i < (BL_TH + 6) * (BL_TW + 6); i += 256) {
        const int r = i / (BL_TW + 6), c = i - r * (BL_TW + 6);
        const int y = reflect101(y0 + r - 3, L.h), x = reflect101(x0 + c - 3, L.w);
        // tiles past the right/bottom edge reflect twice at most for w,h >= 7; clamp defensively
        const int yy = min(max(y, 0), L.h - 1), xx = min(max(x, 0), L.w - 1);
        in[r * (BL_TW + 8) + c] = src[(size_t)yy * L.img_pitch + xx];
    }
    __syncthreads();
    const int k0 = P.gauss_q8[0], k1 = P.gauss_q8[1], k2 = P.gauss_q8[2], k3 = P.gauss_q8[3];
    for (int i = tid; i < (BL_TH + 6) * BL_TW; i += 256) {
        const int r = i / BL_TW, c = i - r * BL_TW;
        const uint8_t *q = &in[r * (BL_TW + 8) + c];
        hz[i] = (uint16_t)(k0 * (q[0] + q[6]) + k1 * (q[1] + q[5]) + k2 * (q[2] + q[4]) + k3 * q[3]);
    }
    __syncthreads();
    uint8_t *dst = L.blur + (size_t)frame * L.blur_frame_stride;
    for (int i = tid; i < BL_TH * BL_TW / 4; i += 256) {
        const int r = i / (BL_TW / 4), c4 = (i - r * (BL_TW / 4)) * 4;
        const int y = y0 + r, x = x0 + c4;
        if (y >= L.h || x >= L.w) continue;
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint16_t *q = &hz[r * BL_TW + c4 + j];
            int s = k0 * (q[0] + q[6 * BL_TW]) + k1 * (q[BL_TW] + q[5 * BL_TW]) + k2 * (q[2 * BL_TW] + q[4 * BL_TW]) + k3 * q[3 * BL_TW];
            s = (s + 32768) >> 16;
            packed |= (uint32_t)min(s, 255) << (8 * j);
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)y * L.blur_pitch + x) = packed;   // pitch % 64 == 0
    }
}

void orb_launch_blur(const OrbParams &P, hipStream_t s)
{
    for (int l = 0; l < P.nlevels; l++) {
        const OrbLevel &L = P.lv[l];
        dim3 grid((L.w + BL_TW - 1) / BL_TW, (L.h + BL_TH - 1) / BL_TH, P.batch);
        hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, s, P, l);
    }
}

// ----------------------------------------------------------------------------------
// A6 + A8  IC_Angle (ORBextractor.cc:75-102) + steered BRIEF (ORBextractor.cc:106-145),
// one wave per keypoint.  Orientation: 2 patch rows per step (lanes 0-30 / 32-62),
// int32 moments, wave reduction.  Descriptor: lane l evaluates tests l, l+64, l+128,
// l+192; a 64-bit ballot is 8 descriptor bytes (LSB-first, as the reference packs them).
// ----------------------------------------------------------------------------------
__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};

// cv::fastAtan2 scalar path, op-by-op, no contraction.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// "orb_sincos" (DESIGN.md): fixed IEEE-double fma sequence, identical to the oracle's.
__device__ __forceinline__ void sincos_det(double x, double *s_out, double *c_out)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const int k = (int)__dadd_rn(__dmul_rn(x, TWO_OVER_PI), 0.5);
    const double dk = (double)k;
    double r = fma(-dk, PIO2_HI, x);
    r = fma(-dk, PIO2_LO, r);
    const double z = __dmul_rn(r, r);
    double ps = fma(z, S6, S5); ps = fma(z, ps, S4); ps = fma(z, ps, S3); ps = fma(z, ps, S2); ps = fma(z, ps, S1);
    const double s = fma(__dmul_rn(r, z), ps, r);
    double pc = fma(z, C6, C5); pc = fma(z, pc, C4); pc = fma(z, pc, C3); pc = fma(z, pc, C2); pc = fma(z, pc, C1);
    const double c = fma(__dmul_rn(z, z), pc, fma(z, -0.5, 1.0));
    switch (k & 3) {
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

__global__ __launch_bounds__(256) void k_orient_desc(OrbParams P)
{
    __shared__ int8_t pat[1024];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    reinterpret_cast<uint32_t *>(pat)[tid] = reinterpret_cast<const uint32_t *>(c_pattern)[tid];
    __syncthreads();
    const int frame = blockIdx.y;
    const int slot = blockIdx.x * 4 + wid;
    if (slot >= P.kps_per_frame) return;
    int lvl = 0;
    for (int l = 1; l < P.nlevels; l++) if (slot >= P.lv[l].kp_base) lvl = l;
    const OrbLevel &L = P.lv[lvl];
    const int idx = slot - L.kp_base;
    if (idx >= P.lvl_count[frame * P.nlevels + lvl]) return;
    const uint32_t key = P.lvl_kp[(size_t)frame * P.kps_per_frame + slot];
    const int x = ORB_KEY_X(key) + ORB_MINB, y = ORB_KEY_Y(key) + ORB_MINB;     // ORBextractor.cc:868-869
    // ---- IC_Angle
    const uint8_t *c = L.img + (size_t)frame * L.img_frame_stride + (size_t)y * L.img_pitch + x;
    const int half = lane >> 5, u = (lane & 31) - ORB_HALF_PATCH;
    int m10 = 0, m01 = 0;
#pragma unroll 4
    for (int it = 0; it < 16; it++) {
        const int v = it - ORB_HALF_PATCH + half * 16;       // half0: -15..0, half1: 1..16
        const int av = v < 0 ? -v : v;
        if (av <= ORB_HALF_PATCH && (lane & 31) < 31) {
            const int d = P.umax[av];
            if (u >= -d && u <= d) {
                const int val = c[v * L.img_pitch + u];
                m10 += u * val;
                m01 += v * val;
            }
        }
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // ---- steered BRIEF on the blurred level
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.f);
    double sd, cd;
    sincos_det((double)__fmul_rn(angle, factor_pi), &sd, &cd);
    const float a = (float)cd, b = (float)sd;
    const uint8_t *bc = L.blur + (size_t)frame * L.blur_frame_stride + (size_t)y * L.blur_pitch + x;
    uint8_t *desc = P.lvl_desc + ((size_t)frame * P.kps_per_frame + slot) * 32;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int8_t *pp = &pat[4 * (lane + 64 * j)];
        const float px0 = (float)pp[0], py0 = (float)pp[1], px1 = (float)pp[2], py1 = (float)pp[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(px0, b), __fmul_rn(py0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(px0, a), __fmul_rn(py0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(px1, b), __fmul_rn(py1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(px1, a), __fmul_rn(py1, b)));
        const int t0 = bc[r0 * L.blur_pitch + c0], t1 = bc[r1 * L.blur_pitch + c1];
        const unsigned long long m = __ballot(t0 < t1);
        if (lane == 0) reinterpret_cast<unsigned long long *>(desc)[j] = m;
    }
    if (lane == 0) P.lvl_angle[(size_t)frame * P.kps_per_frame + slot] = angle;
}

void orb_launch_orient_desc(const OrbParams &P, hipStream_t s)
{
    dim3 grid((P.kps_per_frame + 3) / 4, P.batch);
    hipLaunchKernelGGL(k_orient_desc, grid, dim3(256), 0, s, P);
}

// ----------------------------------------------------------------------------------
// A9  output assembly: level order, pt *= scale for level>0, lapping-area split
// (ORBextractor.cc:1104-1149).  One workgroup per frame.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(OrbParams P)
{
    __shared__ int wsum[8];
    __shared__ int s_cnt[ORB_MAX_LEVELS + 1];
    const int tid = threadIdx.x, frame = blockIdx.x;
    if (tid == 0) {
        int n = 0;
        for (int l = 0; l < P.nlevels; l++) { s_cnt[l] = n; n += P.lvl_count[frame * P.nlevels + l]; }
        s_cnt[P.nlevels] = n;
    }
    __syncthreads();
    const int n = s_cnt[P.nlevels];
    orbhip_keypoint *okp = P.out_kp + (size_t)frame * P.max_kp;
    uint8_t *odesc = P.out_desc + (size_t)frame * P.max_kp * 32;
    int mono_run = 0;
    for (int g0 = 0; g0 < n; g0 += 256) {
        const int g = g0 + tid;
        int lvl = 0, stereo = 0;
        float fx = 0, fy = 0;
        size_t src = 0;
        if (g < n) {
            for (int l = 1; l < P.nlevels; l++) if (g >= s_cnt[l]) lvl = l;
            const OrbLevel &L = P.lv[lvl];
            src = (size_t)frame * P.kps_per_frame + L.kp_base + (g - s_cnt[lvl]);
            const uint32_t key = P.lvl_kp[src];
            fx = (float)(ORB_KEY_X(key) + ORB_MINB);
            fy = (float)(ORB_KEY_Y(key) + ORB_MINB);
            if (lvl != 0) { fx = __fmul_rn(fx, L.scale); fy = __fmul_rn(fy, L.scale); }    // :1131-1133
            stereo = (fx >= (float)P.lap0 && fx <= (float)P.lap1) ? 1 : 0;                 // :1135
        }
        int tot;
        const int mono_before = mono_run + block_excl_scan256((g < n && !stereo) ? 1 : 0, wsum, &tot);
        if (g < n) {
            const int stereo_before = g - mono_before;
            const int slot = stereo ? (n - 1 - stereo_before) : mono_before;
            const OrbLevel &L = P.lv[lvl];
            orbhip_keypoint kp;
            kp.x = fx; kp.y = fy; kp.size = L.size; kp.angle = P.lvl_angle[src];
            kp.response = (float)ORB_KEY_S(P.lvl_kp[src]);
            kp.octave = lvl; kp.class_id = -1;
            okp[slot] = kp;
            const uint4 *d = reinterpret_cast<const uint4 *>(P.lvl_desc + src * 32);
            uint4 *o = reinterpret_cast<uint4 *>(odesc + (size_t)slot * 32);
            o[0] = d[0]; o[1] = d[1];
        }
        mono_run += tot;
    }
    if (tid == 0) { P.out_count[frame] = n; P.out_mono[frame] = mono_run; }
}

void orb_launch_assemble(const OrbParams &P, hipStream_t s)
{
    hipLaunchKernelGGL(k_assemble, dim3(P.batch), dim3(256), 0, s, P);
}
