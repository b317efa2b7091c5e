// LED-to-camera endpoint matching: the image-formation step that follows the two traces of BubbleImageCreator
// (python/BubbleImageCreator.py:278-357, MATLAB kept there as comments).  The reference walks, for every LED ray that
// ends on a light-giving surface, over every camera ray ending on the same surface (O(N_led * N_cam)), keeps the
// nearest one and adds a Gaussian-weighted share of the LED ray's intensity to that camera pixel.  Here the camera
// end points go into a uniform grid of cell size max_dist (counting sort by hashed cell), and every LED ray looks at
// the 27 cells around it: same result, O(N_led * neighbours).
//
// Included at the end of rt_kernels.hip (uses rt_ctx, dev_alloc, CK).  Final-row layout: include/rt_mi355x.h.
#pragma once

namespace {

constexpr int MATCH_MAX_SURF = 64;
constexpr int SCAN_PER_BLOCK = 2048; // 256 threads x 8 counts

struct MatchItem { double x, y, z; int32_t idx, slot; }; // camera end point in bucket order, 32 B

struct MatchPts { // end points of one ray set, SoA
    double* px; double* py; double* pz; double* dx; double* dy; double* dz; double* theta; double* s0; int32_t* slot; int64_t n;
};

struct MatchArgs {
    MatchPts cam, led;
    const MatchItem* items; const uint32_t* offs; uint32_t nb_mask;
    const long long* surf_last; // highest camera index on each listed surface, -1 if none
    double h, inv_h, ang_den, cam_den, led_den, dist_den;
    double* pix; long long* led_cam; double* led_dist; double* led_add;
};

__device__ __forceinline__ int match_cell(double v, double inv_h) {
    double c = floor(v * inv_h);
    c = c > 1e9 ? 1e9 : (c < -1e9 ? -1e9 : c);
    return (int)c;
}
__device__ __forceinline__ uint32_t match_hash(int ix, int iy, int iz, int slot) {
    uint32_t h = (uint32_t)ix * 73856093u ^ (uint32_t)iy * 19349663u ^ (uint32_t)iz * 83492791u ^ (uint32_t)slot * 2654435761u;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
    return h;
}
// |a - b| exactly as norm(c_posn - l_posn) evaluates it for a 3-vector: no fused multiply-adds, left to right
__device__ __forceinline__ double match_dist(double ax, double ay, double az, double bx, double by, double bz) {
#pragma clang fp contract(off)
    const double ex = ax - bx, ey = ay - by, ez = az - bz;
    const double s = (ex * ex + ey * ey) + ez * ez;
    return sqrt(s);
}

// rows -> SoA end points; slot = position of |surface_index| in the list, -1 for rays that never hit, sit on another
// surface or have a non-finite end point.  CAM: also counts the end point into its grid bucket.
template <bool CAM> __global__ void __launch_bounds__(256) k_match_prep(const double* __restrict__ rows, MatchPts o, const int32_t* __restrict__ surf, int n_surf,
                                                                          double inv_h, uint32_t nb_mask, uint32_t* __restrict__ counts, long long* __restrict__ surf_last) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= o.n) return;
    const double* r = rows + i * FIN;
    const double px = r[0], py = r[1], pz = r[2], dx = r[3], dy = r[4], dz = r[5];
    const double nx = r[13], ny = r[14], nz = r[15];
    int slot = -1;
    if (r[19] > 0 && isfinite(px) && isfinite(py) && isfinite(pz)) {
        const double six = fabs(r[16]);
        for (int j = n_surf - 1; j >= 0; j--) if (six == (double)surf[j]) slot = j;
    }
    double ci = -(dx * nx + dy * ny + dz * nz);              // cos_incident, BubbleImageCreator.py:163 (MATLAB columns 1:3)
    ci = ci > 1.0 ? 1.0 : (ci < -1.0 ? -1.0 : ci);           // acos of 1 + ulp is complex in MATLAB; see DESIGN.md
    o.px[i] = px; o.py[i] = py; o.pz[i] = pz; o.dx[i] = dx; o.dy[i] = dy; o.dz[i] = dz; o.theta[i] = acos(ci); o.s0[i] = r[9]; o.slot[i] = slot;
    if (CAM && slot >= 0) {
        atomicAdd(&counts[match_hash(match_cell(px, inv_h), match_cell(py, inv_h), match_cell(pz, inv_h), slot) & nb_mask], 1u);
        atomicMax(&surf_last[slot], (long long)i);
    }
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

__global__ void __launch_bounds__(256) k_match_scatter(MatchPts cam, double inv_h, uint32_t nb_mask, const uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor, MatchItem* __restrict__ items) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cam.n) return;
    const int slot = cam.slot[i];
    if (slot < 0) return;
    const double x = cam.px[i], y = cam.py[i], z = cam.pz[i];
    const uint32_t b = match_hash(match_cell(x, inv_h), match_cell(y, inv_h), match_cell(z, inv_h), slot) & nb_mask;
    const uint32_t at = offs[b] + atomicAdd(&cursor[b], 1u);
    MatchItem it; it.x = x; it.y = y; it.z = z; it.idx = (int32_t)i; it.slot = slot;
    items[at] = it;
}

// One LED ray per lane.  Sequential rule of BubbleImageCreator.py:302-311: the running best is replaced when its
// distance is exactly 0 (the "unset" sentinel) or the new distance is smaller.  That equals: nearest camera ray, lowest
// index on ties - except after a camera ray at distance exactly 0, which the next camera ray on the surface always
// replaces.  With z the highest such index: the answer is (z, 0) if z is the surface's last camera ray, else the
// nearest among the camera rays behind z.  Only candidates closer than max_dist can contribute (:321).
__global__ void __launch_bounds__(256) k_match(const MatchArgs a) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= a.led.n) return;
    const int slot = a.led.slot[l];
    long long best = -1; double bd = 0.0, add = 0.0;
    if (slot >= 0) {
        const double x = a.led.px[l], y = a.led.py[l], z = a.led.pz[l];
        const int cx = match_cell(x, a.inv_h), cy = match_cell(y, a.inv_h), cz = match_cell(z, a.inv_h);
        long long zero = -1;
        for (int pass = 0; pass < 2; pass++) {
            best = -1; bd = 0.0;
            for (int n = 0; n < 27; n++) {
                const int ix = cx + n % 3 - 1, iy = cy + (n / 3) % 3 - 1, iz = cz + n / 9 - 1;
                const uint32_t b = match_hash(ix, iy, iz, slot) & a.nb_mask;
                const uint32_t e = a.offs[b + 1];
                for (uint32_t k = a.offs[b]; k < e; k++) {
                    const MatchItem it = a.items[k];
                    if (it.slot != slot) continue;
                    const double d = match_dist(it.x, it.y, it.z, x, y, z);
                    if (pass == 0 && d == 0.0 && it.idx > zero) zero = it.idx;
                    if (!(d < a.h) || (long long)it.idx <= (pass ? zero : -1LL)) continue;
                    if (best < 0 || d < bd || (d == bd && it.idx < best)) { best = it.idx; bd = d; }
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

    // grid: a power of two of buckets, about two per camera ray
    int64_t nb = 1024; while (nb < 2 * n_cam && nb < ((int64_t)1 << 28)) nb <<= 1;
    const int64_t scan_blocks = (nb + 1 + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    DevBuf b_offs, b_cursor, b_tmp, b_items, b_cpts, b_cslot, b_lpts, b_lslot;
    CK(dev_alloc(ctx, b_offs, sizeof(uint32_t) * (size_t)(nb + 1))); CK(dev_alloc(ctx, b_cursor, sizeof(uint32_t) * (size_t)nb));
    CK(dev_alloc(ctx, b_tmp, sizeof(uint32_t) * (size_t)(scan_blocks + scan_blocks / 1024 + 4096)));
    CK(dev_alloc(ctx, b_items, sizeof(MatchItem) * (size_t)std::max<int64_t>(n_cam, 1)));
    MatchPts cam, led;
    if (int rc = match_alloc_pts(ctx, cam, b_cpts, b_cslot, std::max<int64_t>(n_cam, 1))) return rc;
    if (int rc = match_alloc_pts(ctx, led, b_lpts, b_lslot, n_led)) return rc;
    cam.n = n_cam;
    uint32_t* offs = (uint32_t*)b_offs.p; uint32_t* cursor = (uint32_t*)b_cursor.p;
    CK(hipMemsetAsync(offs, 0, sizeof(uint32_t) * (size_t)(nb + 1), st)); CK(hipMemsetAsync(cursor, 0, sizeof(uint32_t) * (size_t)nb, st));

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    const double h = mp->max_dist, inv_h = 1.0 / h; const uint32_t mask = (uint32_t)(nb - 1);
    if (n_cam) {
        k_match_prep<true><<<(unsigned)((n_cam + 255) / 256), 256, 0, st>>>(d_crow, cam, (const int32_t*)b_surf.p, n_surfaces, inv_h, mask, offs, (long long*)b_last.p);
        CK(hipGetLastError()); ctx->n_launches++;
    }
    k_match_prep<false><<<(unsigned)((n_led + 255) / 256), 256, 0, st>>>(d_lrow, led, (const int32_t*)b_surf.p, n_surfaces, inv_h, mask, nullptr, nullptr);
    CK(hipGetLastError()); ctx->n_launches++;
    if (int rc = match_scan(ctx, offs, nb + 1, (uint32_t*)b_tmp.p)) return rc;
    if (n_cam) {
        k_match_scatter<<<(unsigned)((n_cam + 255) / 256), 256, 0, st>>>(cam, inv_h, mask, offs, cursor, (MatchItem*)b_items.p);
        CK(hipGetLastError()); ctx->n_launches++;
    }
    MatchArgs a;
    a.cam = cam; a.led = led; a.items = (const MatchItem*)b_items.p; a.offs = offs; a.nb_mask = mask; a.surf_last = (const long long*)b_last.p;
    a.h = h; a.inv_h = inv_h; a.ang_den = mp->angle_den; a.cam_den = mp->cam_incidence_den; a.led_den = mp->led_incidence_den; a.dist_den = mp->dist_den;
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
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}
