// LED-to-camera endpoint matching: the image-formation step that follows the two traces of BubbleImageCreator
// (python/BubbleImageCreator.py:278-357, MATLAB kept there as comments).  The reference walks, for every LED ray that
// ends on a light-giving surface, over every camera ray ending on the same surface (O(N_led * N_cam)), keeps the
// nearest one and adds a Gaussian-weighted share of the LED ray's intensity to that camera pixel.  Here the camera
// end points go into a sparse octree (hashed grids of cell size max_dist / 2^level, as deep as the density asks for)
// and every LED ray searches it branch-and-bound from the 27 coarse cells around it: same result, O(N_led log N_cam).
//
// Included at the end of rt_kernels.hip (uses rt_ctx, dev_alloc, CK).  Final-row layout: include/rt_mi355x.h.
#pragma once

namespace {

constexpr int MATCH_MAX_SURF = 64;
constexpr int MATCH_MAX_LEVEL = 10;  // finest cells: max_dist / 2^10
constexpr int MATCH_STACK = 27 + 7 * MATCH_MAX_LEVEL + 9;
constexpr int SCAN_PER_BLOCK = 2048; // 256 threads x 8 counts
constexpr double MATCH_FAR = 1048576.0; // end points beyond 2^20 coarse cells from the origin are refused

struct MatchItem { double x, y, z; int32_t idx, slot; }; // camera end point in leaf-bucket order, 32 B

struct MatchPts { // end points of one ray set, SoA
    double* px; double* py; double* pz; double* dx; double* dy; double* dz; double* theta; double* s0; int32_t* slot; int64_t n;
};

// Sparse octree over the camera end points, held as one hashed table per level: level j has cells of size
// max_dist / 2^j; levels 0..leaf-1 keep one occupancy bit per bucket, the leaf level keeps the end points themselves
// (counting sort by bucket).  A hash collision only makes an empty cell look occupied.
struct MatchGrid {
    const uint32_t* bits[MATCH_MAX_LEVEL + 1]; uint32_t bit_mask[MATCH_MAX_LEVEL + 1]; // per level below the leaf
    const uint32_t* offs; uint32_t leaf_mask; const MatchItem* items; int leaf;
    double h, inv_h;
};

struct MatchArgs {
    MatchPts cam, led; MatchGrid g;
    const long long* surf_last; // highest camera index on each listed surface, -1 if none
    double ang_den, cam_den, led_den, dist_den;
    double* pix; long long* led_cam; double* led_dist; double* led_add;
};

__device__ __forceinline__ uint32_t match_hash(int ix, int iy, int iz, int slot, int level) {
    uint32_t h = (uint32_t)ix * 73856093u ^ (uint32_t)iy * 19349663u ^ (uint32_t)iz * 83492791u ^ (uint32_t)(slot * 16 + level) * 2654435761u;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
    return h;
}
// cell of coordinate t = p / max_dist at a level: exact scaling by 2^level keeps the levels nested
__device__ __forceinline__ int match_cell(double t, int level) { return (int)floor(t * (double)(1 << level)); }
// |a - b| exactly as norm(c_posn - l_posn) evaluates it for a 3-vector: no fused multiply-adds, left to right
__device__ __forceinline__ double match_dist(double ax, double ay, double az, double bx, double by, double bz) {
#pragma clang fp contract(off)
    const double ex = ax - bx, ey = ay - by, ez = az - bz;
    const double s = (ex * ex + ey * ey) + ez * ez;
    return sqrt(s);
}

// rows -> SoA end points; slot = position of |surface_index| in the list, -1 for rays that never hit, sit on another
// surface or have a non-finite end point.  CAM: also marks the coarse cell and counts the rays taking part.
// stats: [0] camera rays taking part [1] end points too far out [2] occupied coarse cells (k_match_popcount)
template <bool CAM> __global__ void __launch_bounds__(256) k_match_prep(const double* __restrict__ rows, MatchPts o, const int32_t* __restrict__ surf, int n_surf,
                                                                          double inv_h, uint32_t* __restrict__ bits0, uint32_t mask0, long long* __restrict__ surf_last,
                                                                          unsigned long long* __restrict__ stats) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int slot = -1; bool far = false;
    double px = 0, py = 0, pz = 0;
    if (i < o.n) {
        const double* r = rows + i * FIN;
        px = r[0]; py = r[1]; pz = r[2];
        const double dx = r[3], dy = r[4], dz = r[5], nx = r[13], ny = r[14], nz = r[15];
        if (r[19] > 0 && isfinite(px) && isfinite(py) && isfinite(pz)) {
            const double six = fabs(r[16]);
            for (int j = n_surf - 1; j >= 0; j--) if (six == (double)surf[j]) slot = j;
            if (slot >= 0 && !(fabs(px * inv_h) < MATCH_FAR && fabs(py * inv_h) < MATCH_FAR && fabs(pz * inv_h) < MATCH_FAR)) { slot = -1; far = true; }
        }
        double ci = -(dx * nx + dy * ny + dz * nz);              // cos_incident, BubbleImageCreator.py:163 (MATLAB columns 1:3)
        ci = ci > 1.0 ? 1.0 : (ci < -1.0 ? -1.0 : ci);           // acos of 1 + ulp is complex in MATLAB; see DESIGN.md
        o.px[i] = px; o.py[i] = py; o.pz[i] = pz; o.dx[i] = dx; o.dy[i] = dy; o.dz[i] = dz; o.theta[i] = acos(ci); o.s0[i] = r[9]; o.slot[i] = slot;
    }
    if (far) atomicAdd(&stats[1], 1ull);
    if (CAM) {
        if (slot >= 0) {
            const uint32_t b = match_hash(match_cell(px * inv_h, 0), match_cell(py * inv_h, 0), match_cell(pz * inv_h, 0), slot, 0) & mask0;
            if (!(bits0[b >> 5] >> (b & 31) & 1u)) atomicOr(&bits0[b >> 5], 1u << (b & 31));
        }
        const unsigned long long on = __ballot(slot >= 0);
        for (unsigned long long todo = on; todo;) {              // one atomic per wave and surface: the highest lane has the highest row
            const int s = __shfl(slot, __ffsll((long long)todo) - 1);
            const unsigned long long same = __ballot(slot == s) & todo;
            if ((int)(threadIdx.x & 63) == 63 - __clzll((long long)same)) atomicMax(&surf_last[s], (long long)i);
            todo &= ~same;
        }
        if ((threadIdx.x & 63) == 0 && on) atomicAdd(&stats[0], (unsigned long long)__popcll(on));
    }
}

__global__ void __launch_bounds__(256) k_match_popcount(const uint32_t* __restrict__ bits, int64_t words, unsigned long long* __restrict__ out) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) c += (unsigned)__popc(bits[i]);
    for (int s = 32; s > 0; s >>= 1) c += __shfl_down(c, s);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// occupancy bits of levels 1..leaf-1 and the leaf counts
__global__ void __launch_bounds__(256) k_match_insert(MatchPts cam, MatchGrid g, uint32_t* const* __restrict__ bits_rw, uint32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cam.n) return;
    const int slot = cam.slot[i];
    if (slot < 0) return;
    const double tx = cam.px[i] * g.inv_h, ty = cam.py[i] * g.inv_h, tz = cam.pz[i] * g.inv_h;
    for (int j = 1; j < g.leaf; j++) {
        const uint32_t b = match_hash(match_cell(tx, j), match_cell(ty, j), match_cell(tz, j), slot, j) & g.bit_mask[j];
        uint32_t* w = bits_rw[j] + (b >> 5);
        if (!(*w >> (b & 31) & 1u)) atomicOr(w, 1u << (b & 31));
    }
    atomicAdd(&counts[match_hash(match_cell(tx, g.leaf), match_cell(ty, g.leaf), match_cell(tz, g.leaf), slot, g.leaf) & g.leaf_mask], 1u);
}

// exclusive prefix sum of 32-bit counts, SCAN_PER_BLOCK per block.  APPLY = false: block totals only.
template <bool APPLY> __global__ void __launch_bounds__(256) k_match_scan(uint32_t* __restrict__ data, int64_t n, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t part[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_PER_BLOCK + (int64_t)threadIdx.x * 8;
    uint32_t v[8]; uint32_t t = 0;
    for (int k = 0; k < 8; k++) { v[k] = base + k < n ? data[base + k] : 0u; t += v[k]; }
    part[threadIdx.x] = t; __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)s ? part[threadIdx.x - s] : 0u; __syncthreads();
        part[threadIdx.x] += add; __syncthreads();
    }
    if (!APPLY) { if (threadIdx.x == 255) bsum[blockIdx.x] = part[255]; return; }
    uint32_t run = part[threadIdx.x] - t + (bsum ? bsum[blockIdx.x] : 0u);
    for (int k = 0; k < 8; k++) { if (base + k < n) data[base + k] = run; run += v[k]; }
}

__global__ void __launch_bounds__(256) k_match_scatter(MatchPts cam, MatchGrid g, uint32_t* __restrict__ cursor, MatchItem* __restrict__ items) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cam.n) return;
    const int slot = cam.slot[i];
    if (slot < 0) return;
    const double x = cam.px[i], y = cam.py[i], z = cam.pz[i];
    const uint32_t b = match_hash(match_cell(x * g.inv_h, g.leaf), match_cell(y * g.inv_h, g.leaf), match_cell(z * g.inv_h, g.leaf), slot, g.leaf) & g.leaf_mask;
    const uint32_t at = g.offs[b] + atomicAdd(&cursor[b], 1u);
    MatchItem it; it.x = x; it.y = y; it.z = z; it.idx = (int32_t)i; it.slot = slot;
    items[at] = it;
}

__device__ __forceinline__ bool match_occupied(const MatchGrid& g, int ix, int iy, int iz, int slot, int level) {
    if (level == g.leaf) { const uint32_t b = match_hash(ix, iy, iz, slot, level) & g.leaf_mask; return g.offs[b + 1] > g.offs[b]; }
    const uint32_t b = match_hash(ix, iy, iz, slot, level) & g.bit_mask[level];
    return g.bits[level][b >> 5] >> (b & 31) & 1u;
}
// lower bound of the distance from t (in units of max_dist) to anything in a cell, a little short of the true one so
// that end points an ulp outside their cell's box are never pruned
__device__ __forceinline__ double match_box_dist(double tx, double ty, double tz, int ix, int iy, int iz, int level) {
    const double w = 1.0 / (double)(1 << level);
    const double ax = fmax(fmax(ix * w - tx, tx - (ix + 1) * w), 0.0), ay = fmax(fmax(iy * w - ty, ty - (iy + 1) * w), 0.0), az = fmax(fmax(iz * w - tz, tz - (iz + 1) * w), 0.0);
    return sqrt(ax * ax + ay * ay + az * az) * (1.0 - 1e-9);
}

// One LED ray per lane.  Sequential rule of BubbleImageCreator.py:302-311: the running best is replaced when its
// distance is exactly 0 (the "unset" sentinel) or the new distance is smaller.  That equals: nearest camera ray, lowest
// index on ties - except after a camera ray at distance exactly 0, which the next camera ray on the surface always
// replaces.  With z the highest such index: the answer is (z, 0) if z is the surface's last camera ray, else the
// nearest among the camera rays behind z.  Only candidates closer than max_dist can contribute (:321).
// The search is branch and bound down the octree, nearest child first.
__global__ void __launch_bounds__(256) k_match(const MatchArgs a) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= a.led.n) return;
    const MatchGrid& g = a.g;
    const int slot = a.led.slot[l];
    long long best = -1; double bd = 0.0, add = 0.0;
    if (slot >= 0) {
        const double x = a.led.px[l], y = a.led.py[l], z = a.led.pz[l];
        const double tx = x * g.inv_h, ty = y * g.inv_h, tz = z * g.inv_h;
        const int cx = match_cell(tx, 0), cy = match_cell(ty, 0), cz = match_cell(tz, 0);
        long long zero = -1;
        int4 stack[MATCH_STACK];                               // (ix, iy, iz, level); occupancy is looked up when a cell is popped
        for (int pass = 0; pass < 2; pass++) {
            best = -1; bd = g.h;                               // bound: only distances < max_dist count
            double bt = 1.0;                                   // the same in units of max_dist, for the boxes
            int sp = 0;
            for (int n = 0; n < 27; n++) {                     // own cell (13) is pushed last, so it is looked at first
                const int m = n == 13 ? 26 : (n == 26 ? 13 : n);
                stack[sp++] = make_int4(cx + m % 3 - 1, cy + (m / 3) % 3 - 1, cz + m / 9 - 1, 0);
            }
            while (sp > 0) {
                const int4 c4 = stack[--sp];
                const int ix = c4.x, iy = c4.y, iz = c4.z, lev = c4.w;
                if (match_box_dist(tx, ty, tz, ix, iy, iz, lev) > bt) continue;
                if (lev == g.leaf) {
                    const uint32_t b = match_hash(ix, iy, iz, slot, lev) & g.leaf_mask;
                    const uint32_t e = g.offs[b + 1];
                    for (uint32_t k = g.offs[b]; k < e; k++) {
                        const MatchItem it = g.items[k];
                        if (it.slot != slot) continue;
                        const double d = match_dist(it.x, it.y, it.z, x, y, z);
                        if (pass == 0 && d == 0.0 && it.idx > zero) zero = it.idx;
                        if (!(d < g.h) || (long long)it.idx <= (pass ? zero : -1LL)) continue;
                        if (best < 0 || d < bd || (d == bd && it.idx < best)) { best = it.idx; bd = d; bt = d * g.inv_h * (1.0 + 1e-12); }
                    }
                    continue;
                }
                if (!match_occupied(g, ix, iy, iz, slot, lev)) continue;
                // children, the one nearest to the LED ray's end point on top of the stack
                const int nx = tx * (double)(2 << lev) >= (double)(2 * ix + 1), ny = ty * (double)(2 << lev) >= (double)(2 * iy + 1), nz = tz * (double)(2 << lev) >= (double)(2 * iz + 1);
                for (int c = 7; c >= 0; c--) {                 // c = 0: the nearest octant
                    const int jx = 2 * ix + ((c & 1) ^ nx), jy = 2 * iy + (((c >> 1) & 1) ^ ny), jz = 2 * iz + (((c >> 2) & 1) ^ nz);
                    if (c && match_box_dist(tx, ty, tz, jx, jy, jz, lev + 1) > bt) continue;
                    stack[sp++] = make_int4(jx, jy, jz, lev + 1);
                }
            }
            if (zero < 0) break;                                         // the ordinary case: one pass
            if (pass == 0 && zero == a.surf_last[slot]) { best = zero; bd = 0.0; break; }
        }
        if (best >= 0) {
            const double cxd = a.cam.dx[best], cyd = a.cam.dy[best], czd = a.cam.dz[best];
            const double lx = a.led.dx[l], ly = a.led.dy[l], lz = a.led.dz[l];
            double cb = (cxd * lx + cyd * ly + czd * lz) / (sqrt(cxd * cxd + cyd * cyd + czd * czd) * sqrt(lx * lx + ly * ly + lz * lz));
            cb = cb > 1.0 ? 1.0 : (cb < -1.0 ? -1.0 : cb);
            const double ang = acos(cb), ci = a.cam.theta[best], li = a.led.theta[l];
            add = a.led.s0[l] * exp(-(ang * ang) / a.ang_den - (ci * ci) / a.cam_den - (bd * bd) / a.dist_den - (li * li) / a.led_den); // :338
            atomicAdd(&a.pix[best], add);
        }
    }
    if (a.led_cam) a.led_cam[l] = best;
    if (a.led_dist) a.led_dist[l] = best >= 0 ? bd : 0.0;
    if (a.led_add) a.led_add[l] = add;
}

int match_scan(rt_ctx* ctx, uint32_t* d, int64_t n, uint32_t* tmp) {
    const int64_t blocks = (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    if (blocks == 1) { k_match_scan<true><<<1, 256, 0, ctx->stream>>>(d, n, nullptr); CK(hipGetLastError()); return 0; }
    k_match_scan<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d, n, tmp); CK(hipGetLastError());
    if (int rc = match_scan(ctx, tmp, blocks, tmp + ((blocks + 255) & ~(int64_t)255))) return rc;
    k_match_scan<true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d, n, tmp); CK(hipGetLastError());
    return 0;
}

int match_alloc_pts(rt_ctx* ctx, MatchPts& p, DevBuf& bd, DevBuf& bs, int64_t n) {
    CK(dev_alloc(ctx, bd, sizeof(double) * 8 * (size_t)n)); CK(dev_alloc(ctx, bs, sizeof(int32_t) * (size_t)n));
    double* b = (double*)bd.p; const size_t s = (size_t)n;
    p.px = b; p.py = b + s; p.pz = b + 2 * s; p.dx = b + 3 * s; p.dy = b + 4 * s; p.dz = b + 5 * s; p.theta = b + 6 * s; p.s0 = b + 7 * s;
    p.slot = (int32_t*)bs.p; p.n = n;
    return 0;
}

} // namespace

extern "C" int rt_match_endpoints(rt_ctx* ctx, const double* cam_rows, int64_t n_cam, const double* led_rows, int64_t n_led, const int32_t* surfaces, int32_t n_surfaces,
                                  const rt_match_params* mp, double* pixel_intensity, int64_t* led_cam, double* led_dist, double* led_add) {
    if (!ctx) return -1;
    if (n_cam < 0 || n_led < 0 || n_cam > 0x7fffffffLL || n_led > ((int64_t)1 << 40)) { ctx->err = "rt_match_endpoints: ray counts out of range (camera rays < 2^31)"; return -1; }
    if (n_surfaces < 0 || n_surfaces > MATCH_MAX_SURF) { ctx->err = "rt_match_endpoints: at most 64 surfaces"; return -1; }
    if (!mp || !(mp->max_dist > 0) || !std::isfinite(mp->max_dist)) { ctx->err = "rt_match_endpoints: max_dist must be positive and finite"; return -1; }
    if ((n_cam && (!cam_rows || !pixel_intensity)) || (n_led && !led_rows) || (n_surfaces && !surfaces)) { ctx->err = "rt_match_endpoints: null array"; return -1; }
    CK(hipSetDevice(ctx->device));
    const bool dev = mp->flags & RT_MATCH_DEVICE_POINTERS;
    ctx->kernel_ms = 0; ctx->n_launches = 0;
    hipStream_t st = ctx->stream;
    const size_t rowb = sizeof(double) * FIN;

    DevBuf b_pix, b_lcam, b_ldist, b_ladd;
    double* d_pix = pixel_intensity;
    if (!dev && n_cam) { CK(dev_alloc(ctx, b_pix, sizeof(double) * (size_t)n_cam)); d_pix = (double*)b_pix.p; }
    if (n_cam) CK(hipMemsetAsync(d_pix, 0, sizeof(double) * (size_t)n_cam, st));
    long long* d_lcam = (long long*)led_cam; double* d_ldist = led_dist; double* d_ladd = led_add;
    if (!dev && n_led) {
        if (led_cam) { CK(dev_alloc(ctx, b_lcam, 8 * (size_t)n_led)); d_lcam = (long long*)b_lcam.p; }
        if (led_dist) { CK(dev_alloc(ctx, b_ldist, 8 * (size_t)n_led)); d_ldist = (double*)b_ldist.p; }
        if (led_add) { CK(dev_alloc(ctx, b_ladd, 8 * (size_t)n_led)); d_ladd = (double*)b_ladd.p; }
    }
    if (n_led == 0) { // nothing shines: the zeros are the answer
        if (!dev && n_cam) { CK(hipMemcpyAsync(pixel_intensity, d_pix, sizeof(double) * (size_t)n_cam, hipMemcpyDeviceToHost, st)); }
        CK(hipStreamSynchronize(st));
        return 0;
    }

    // inputs on the device
    DevBuf b_crow, b_lrow, b_surf, b_last;
    const double* d_crow = cam_rows; const double* d_lrow = led_rows;
    if (!dev) {
        if (n_cam) { CK(dev_alloc(ctx, b_crow, rowb * (size_t)n_cam)); CK(hipMemcpyAsync(b_crow.p, cam_rows, rowb * (size_t)n_cam, hipMemcpyHostToDevice, st)); d_crow = (const double*)b_crow.p; }
        CK(dev_alloc(ctx, b_lrow, rowb * (size_t)n_led)); CK(hipMemcpyAsync(b_lrow.p, led_rows, rowb * (size_t)n_led, hipMemcpyHostToDevice, st)); d_lrow = (const double*)b_lrow.p;
    }
    CK(dev_alloc(ctx, b_surf, sizeof(int32_t) * MATCH_MAX_SURF)); CK(dev_alloc(ctx, b_last, sizeof(long long) * MATCH_MAX_SURF));
    if (n_surfaces) CK(hipMemcpyAsync(b_surf.p, surfaces, sizeof(int32_t) * (size_t)n_surfaces, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(b_last.p, 0xff, sizeof(long long) * MATCH_MAX_SURF, st)); // -1

    // coarse level: which cells of size max_dist hold camera end points, how many rays take part
    const double h = mp->max_dist, inv_h = 1.0 / h;
    const uint32_t mask0 = (1u << 22) - 1;
    DevBuf b_bits0, b_stats, b_cpts, b_cslot, b_lpts, b_lslot;
    CK(dev_alloc(ctx, b_bits0, (size_t)(mask0 + 1) / 8)); CK(dev_alloc(ctx, b_stats, 64));
    CK(hipMemsetAsync(b_bits0.p, 0, (size_t)(mask0 + 1) / 8, st)); CK(hipMemsetAsync(b_stats.p, 0, 64, st));
    MatchPts cam, led;
    if (int rc = match_alloc_pts(ctx, cam, b_cpts, b_cslot, std::max<int64_t>(n_cam, 1))) return rc;
    if (int rc = match_alloc_pts(ctx, led, b_lpts, b_lslot, n_led)) return rc;
    cam.n = n_cam;
    hipEvent_t e0 = ctx->ev[0], e1 = ctx->ev[1];
    CK(hipEventRecord(e0, st));
    unsigned long long* d_stats = (unsigned long long*)b_stats.p;
    if (n_cam) {
        k_match_prep<true><<<(unsigned)((n_cam + 255) / 256), 256, 0, st>>>(d_crow, cam, (const int32_t*)b_surf.p, n_surfaces, inv_h, (uint32_t*)b_bits0.p, mask0, (long long*)b_last.p, d_stats);
        CK(hipGetLastError()); ctx->n_launches++;
    }
    k_match_prep<false><<<(unsigned)((n_led + 255) / 256), 256, 0, st>>>(d_lrow, led, (const int32_t*)b_surf.p, n_surfaces, inv_h, nullptr, 0, nullptr, d_stats);
    CK(hipGetLastError()); ctx->n_launches++;
    k_match_popcount<<<256, 256, 0, st>>>((const uint32_t*)b_bits0.p, (int64_t)(mask0 + 1) / 32, d_stats + 2);
    CK(hipGetLastError()); ctx->n_launches++;
    unsigned long long stats[3];
    CK(hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    if (stats[1]) { ctx->err = "rt_match_endpoints: end points further than 2^20 * max_dist from the origin"; return -1; }
    const int64_t n_on = (int64_t)stats[0]; const double occ0 = (double)std::max<unsigned long long>(stats[2], 1);
    // leaf level: end points lie on surfaces, so a cell's load falls by about 4 per level; aim at a handful per leaf
    int leaf = 0;
    while (leaf < ctx->match_max_level && (double)n_on / (occ0 * std::pow(4.0, leaf)) > 6.0) leaf++;
    MatchGrid g; memset(&g, 0, sizeof(g));
    g.h = h; g.inv_h = inv_h; g.leaf = leaf;
    size_t bit_words[MATCH_MAX_LEVEL + 1] = { 0 }, bit_total = 0;
    for (int j = 1; j < leaf; j++) {
        uint64_t nbits = 1u << 16; const double want = 16.0 * occ0 * std::pow(4.0, j);
        while ((double)nbits < want && nbits < (1ull << 31)) nbits <<= 1;
        g.bit_mask[j] = (uint32_t)(nbits - 1); bit_words[j] = (size_t)(nbits / 32); bit_total += bit_words[j];
    }
    int64_t nb = 1024; while (nb < 2 * n_on && nb < ((int64_t)1 << 28)) nb <<= 1;
    g.leaf_mask = (uint32_t)(nb - 1);
    const int64_t scan_blocks = (nb + 1 + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    DevBuf b_bits, b_ptrs, b_offs, b_cursor, b_tmp, b_items;
    CK(dev_alloc(ctx, b_bits, 4 * std::max<size_t>(bit_total, 1))); CK(dev_alloc(ctx, b_ptrs, sizeof(uint32_t*) * (MATCH_MAX_LEVEL + 1)));
    CK(dev_alloc(ctx, b_offs, sizeof(uint32_t) * (size_t)(nb + 1))); CK(dev_alloc(ctx, b_cursor, sizeof(uint32_t) * (size_t)nb));
    CK(dev_alloc(ctx, b_tmp, sizeof(uint32_t) * (size_t)(scan_blocks + scan_blocks / 1024 + 4096)));
    CK(dev_alloc(ctx, b_items, sizeof(MatchItem) * (size_t)std::max<int64_t>(n_on, 1)));
    uint32_t* h_ptrs[MATCH_MAX_LEVEL + 1] = { nullptr };
    { size_t at = 0; for (int j = 1; j < leaf; j++) { h_ptrs[j] = (uint32_t*)b_bits.p + at; at += bit_words[j]; } }
    h_ptrs[0] = (uint32_t*)b_bits0.p; g.bit_mask[0] = mask0;
    for (int j = 0; j <= MATCH_MAX_LEVEL; j++) g.bits[j] = h_ptrs[j];
    uint32_t* offs = (uint32_t*)b_offs.p; uint32_t* cursor = (uint32_t*)b_cursor.p;
    g.offs = offs; g.items = (const MatchItem*)b_items.p;
    CK(hipMemcpyAsync(b_ptrs.p, h_ptrs, sizeof(h_ptrs), hipMemcpyHostToDevice, st));
    if (bit_total) CK(hipMemsetAsync(b_bits.p, 0, 4 * bit_total, st));
    CK(hipMemsetAsync(offs, 0, sizeof(uint32_t) * (size_t)(nb + 1), st)); CK(hipMemsetAsync(cursor, 0, sizeof(uint32_t) * (size_t)nb, st));
    if (n_cam) {
        k_match_insert<<<(unsigned)((n_cam + 255) / 256), 256, 0, st>>>(cam, g, (uint32_t* const*)b_ptrs.p, offs);
        CK(hipGetLastError()); ctx->n_launches++;
    }
    if (int rc = match_scan(ctx, offs, nb + 1, (uint32_t*)b_tmp.p)) return rc;
    if (n_cam) {
        k_match_scatter<<<(unsigned)((n_cam + 255) / 256), 256, 0, st>>>(cam, g, cursor, (MatchItem*)b_items.p);
        CK(hipGetLastError()); ctx->n_launches++;
    }
    MatchArgs a;
    a.cam = cam; a.led = led; a.g = g; a.surf_last = (const long long*)b_last.p;
    a.ang_den = mp->angle_den; a.cam_den = mp->cam_incidence_den; a.led_den = mp->led_incidence_den; a.dist_den = mp->dist_den;
    a.pix = d_pix; a.led_cam = d_lcam; a.led_dist = d_ldist; a.led_add = d_ladd;
    k_match<<<(unsigned)((n_led + 255) / 256), 256, 0, st>>>(a);
    CK(hipGetLastError()); ctx->n_launches++;
    CK(hipEventRecord(e1, st));
    if (!dev) {
        if (n_cam) CK(hipMemcpyAsync(pixel_intensity, d_pix, sizeof(double) * (size_t)n_cam, hipMemcpyDeviceToHost, st));
        if (led_cam) CK(hipMemcpyAsync(led_cam, d_lcam, 8 * (size_t)n_led, hipMemcpyDeviceToHost, st));
        if (led_dist) CK(hipMemcpyAsync(led_dist, d_ldist, 8 * (size_t)n_led, hipMemcpyDeviceToHost, st));
        if (led_add) CK(hipMemcpyAsync(led_add, d_ladd, 8 * (size_t)n_led, hipMemcpyDeviceToHost, st));
    }
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ctx->kernel_ms = ms;
    return 0;
}
