// tools/mb_queue.hip -- streaming microbenchmark for the layout of the tally-only ray queue (96 B per ray).
// Copies N records from one queue to another, the way k_level_phys does (2 blocks of 256 per CU, one record per
// lane in flight, through a list of slots that leaves out every `skip`-th ray), in three layouts:
//   soa   : 11 planes of doubles + 2 planes of words, 8 B / 4 B per lane and instruction (round 2's layout)
//   aos   : 96-byte rows, lane = row, six 16-byte loads / stores per lane at a stride of 96 B
//   coop  : 96-byte rows, six lanes per row: every instruction of a wave moves ten whole rows and two thirds (contiguous where
//           the rows are neighbours), transposed through LDS into one row per lane
// build: hipcc --offload-arch=gfx950 -O3 -o build/mb_queue tools/mb_queue.hip ; run: build/mb_queue [rays] [skip]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BLOCK = 256;
typedef double __attribute__((ext_vector_type(2))) d2;

__global__ void __launch_bounds__(BLOCK, 2) k_soa(const double* __restrict__ in_d, const int32_t* __restrict__ in_m, double* __restrict__ out_d, int32_t* __restrict__ out_m,
                                                  const uint32_t* __restrict__ list, int64_t st, int64_t seg_cap, const uint32_t* __restrict__ seg_n) {
    const int64_t seg0 = (int64_t)blockIdx.x * seg_cap, n = seg0 + seg_n[blockIdx.x];
    for (int64_t base = seg0 + threadIdx.x; base < n; base += BLOCK) {
        const int64_t idx = seg0 + list[base];
        double v[11]; int32_t a, b;
#pragma unroll
        for (int f = 0; f < 11; f++) v[f] = in_d[f * st + idx];
        a = in_m[idx]; b = in_m[2 * st + idx];
#pragma unroll
        for (int f = 0; f < 11; f++) out_d[f * st + base] = v[f] + 1.0;
        out_m[base] = a; out_m[2 * st + base] = b + 1;
    }
}
__global__ void __launch_bounds__(BLOCK, 2) k_aos(const d2* __restrict__ in, d2* __restrict__ out, const uint32_t* __restrict__ list, int64_t seg_cap, const uint32_t* __restrict__ seg_n) {
    const int64_t seg0 = (int64_t)blockIdx.x * seg_cap, n = seg0 + seg_n[blockIdx.x];
    for (int64_t base = seg0 + threadIdx.x; base < n; base += BLOCK) {
        const int64_t idx = seg0 + list[base];
        d2 v[6];
#pragma unroll
        for (int f = 0; f < 6; f++) v[f] = in[idx * 6 + f];
#pragma unroll
        for (int f = 0; f < 6; f++) { v[f].x += 1.0; out[base * 6 + f] = v[f]; }
    }
}
// six lanes per row.  A wave handles 64 rows per iteration in six instructions: instruction j, lane t moves piece (64 j + t) of the
// 384 pieces (16 B) of the wave's 64 rows: row (64 j + t) / 6, piece (64 j + t) % 6.  LDS image per wave: 64 rows x 96 B.
__global__ void __launch_bounds__(BLOCK, 2) k_coop(const d2* __restrict__ in, d2* __restrict__ out, const uint32_t* __restrict__ list, int64_t seg_cap, const uint32_t* __restrict__ seg_n) {
    __shared__ d2 lds[BLOCK / 64][64 * 6 + 8];
    const int64_t seg0 = (int64_t)blockIdx.x * seg_cap, n = seg0 + seg_n[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    d2* L = lds[w];
    for (int64_t wbase = seg0 + (threadIdx.x & ~63); wbase < n; wbase += BLOCK) {
        const int64_t li = wbase + lane;
        const uint32_t my = li < n ? list[li] : 0xFFFFFFFFu; // slot of the row this lane will own
        d2 v[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int pc = 64 * j + lane, r = pc / 6, c = pc - 6 * r;
            const uint32_t slot = (uint32_t)__shfl((int)my, r);
            v[j] = slot != 0xFFFFFFFFu ? in[(seg0 + slot) * 6 + c] : d2{ 0, 0 };
        }
#pragma unroll
        for (int j = 0; j < 6; j++) L[64 * j + lane] = v[j]; // piece index = row * 6 + c
        __builtin_amdgcn_wave_barrier();
        d2 r[6];
#pragma unroll
        for (int f = 0; f < 6; f++) { r[f] = L[lane * 6 + f]; r[f].x += 1.0; }
        __builtin_amdgcn_wave_barrier();
        // back the same way (output rows are consecutive: wbase + lane)
#pragma unroll
        for (int f = 0; f < 6; f++) L[lane * 6 + f] = r[f];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int pc = 64 * j + lane, rr = pc / 6;
            if (wbase + rr < n) out[wbase * 6 + pc] = L[pc];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : (int64_t)1 << 24;
    const int skip = argc > 2 ? atoi(argv[2]) : 4; // every skip-th ray is left out of the list (0: none)
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int nseg = pr.multiProcessorCount * 2;
    const int64_t seg_cap = ((n + nseg - 1) / nseg + 63) & ~(int64_t)63, cap = seg_cap * nseg;
    std::vector<uint32_t> list((size_t)cap, 0), segn((size_t)nseg, 0);
    int64_t listed = 0;
    for (int s = 0; s < nseg; s++) {
        uint32_t k = 0;
        for (int64_t j = 0; j < seg_cap && s * seg_cap + j < n; j++) if (skip == 0 || (j % skip) != skip - 1) list[(size_t)(s * seg_cap + k++)] = (uint32_t)j;
        segn[s] = k; listed += k;
    }
    double *a, *b; uint32_t *dl, *dn; int32_t *ma, *mb;
    CK(hipMalloc(&a, 96 * (size_t)cap)); CK(hipMalloc(&b, 96 * (size_t)cap)); CK(hipMalloc(&ma, 12 * (size_t)cap)); CK(hipMalloc(&mb, 12 * (size_t)cap));
    CK(hipMalloc(&dl, 4 * (size_t)cap)); CK(hipMalloc(&dn, 4 * (size_t)nseg));
    CK(hipMemset(a, 0, 96 * (size_t)cap)); CK(hipMemset(b, 0, 96 * (size_t)cap)); CK(hipMemset(ma, 0, 12 * (size_t)cap)); CK(hipMemset(mb, 0, 12 * (size_t)cap));
    CK(hipMemcpy(dl, list.data(), 4 * (size_t)cap, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, segn.data(), 4 * (size_t)nseg, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = { "soa", "aos", "coop" };
    for (int v = 0; v < 3; v++) {
        float best = 1e30f;
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k_soa, dim3(nseg), dim3(BLOCK), 0, 0, a, ma, b, mb, dl, cap, seg_cap, dn);
            else if (v == 1) hipLaunchKernelGGL(k_aos, dim3(nseg), dim3(BLOCK), 0, 0, (const d2*)a, (d2*)b, dl, seg_cap, dn);
            else hipLaunchKernelGGL(k_coop, dim3(nseg), dim3(BLOCK), 0, 0, (const d2*)a, (d2*)b, dl, seg_cap, dn);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it > 0 && ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-5s rays %lld listed %lld skip %d: %.3f ms  %.2f TB/s of listed bytes (96 B read + 96 B written per listed ray)\n", names[v], (long long)n, (long long)listed, skip, best,
               192.0 * (double)listed / (best * 1e-3) / 1e12);
    }
    return 0;
}
