// hostsim.cpp -- TEST AID: compiles raytracer_amd/csrc/rt_device.hpp (the text the HIP kernels inline) for the host so
// that the per-ray arithmetic can be compared with the oracle on a machine without a GPU.  Level-synchronous driver
// in the reference's order.  Not part of the product: nothing in raytracer_amd/ loads it.
#define RT_HD static inline
#define RT_ANY(x) (x)
#include "../../raytracer_amd/csrc/rt_device.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace rt;

struct Params { int32_t max_scatters, flags, compat, unified_maxiter; double min_travel_length, follow_threshold[2], tir_handling; uint64_t seed; int64_t ray_id_base; };
struct Result { int K, S, num_steps; std::vector<std::vector<double>> rec; std::vector<double> table; std::vector<int64_t> live, scat; std::vector<double> fin; };

static void tally(std::vector<double>& t, int S, int k, int cat, int smix, int base, double v) {
    if (smix == 0) { if (base == 0 && S > 0) { t[((size_t)(k * 5 + cat) * S) * 2] += v; t[((size_t)(k * 5 + cat) * S) * 2 + 1] += v; } return; }
    int idx = std::abs(smix) - base;
    if (idx >= 0 && idx < S) t[((size_t)(k * 5 + cat) * S + idx) * 2 + (smix > 0 ? 0 : 1)] += v;
}

extern "C" int hs_trace(const double* sp, const double* rays, int64_t n, const double* table, int S, const double* atoms, const uint32_t* code,
                        const double* lut, int lut_n, const Params* prm, Result** out) {
    int nA = 0; // atoms referenced by the programs
    for (int si = 0; si < S; si++) { int off = (int)table[(size_t)si * ROW + 27], len = (int)table[(size_t)si * ROW + 28]; for (int i = off; i < off + len; i++) if ((code[i] >> 24) == 1) nA = std::max(nA, (int)(code[i] & 0xFFFFFFu) + 1); }
    std::vector<double> recs; std::vector<uint32_t> smeta;
    build_device_tables(table, S, atoms, nA, code, recs, smeta);
    std::vector<double> prog;
    const char* nf = getenv("RT_NO_FRAMES");
    build_program(table, S, atoms, code, !(prm->compat & COMPAT_PY_ROOT2) && !(nf && atoi(nf) > 0), recs, smeta, prog);
    finish_device_tables(recs);
    Geom G{ table, recs.data(), smeta.data(), table, S, lut, lut_n - 1, nullptr, prog.data(), (int)(prog.size() / PREC) - 1 };
    Prm P; P.K = prm->max_scatters; P.flags = prm->flags; P.compat = prm->compat; P.maxiter = prm->unified_maxiter > 0 ? prm->unified_maxiter : 10000;
    P.base = (prm->compat & COMPAT_PY_INDEX) ? 0 : 1; P.min_travel = prm->min_travel_length; P.thrA = prm->follow_threshold[0]; P.thrB = prm->follow_threshold[1];
    P.tir = prm->tir_handling; P.key0 = (uint32_t)prm->seed; P.key1 = (uint32_t)(prm->seed >> 32);
    Result* R = new Result; R->K = P.K; R->S = S; R->rec.resize(P.K); R->table.assign((size_t)P.K * 5 * S * 2, 0.0); R->live.assign(P.K, 0); R->scat.assign(P.K, 0);
    bool wantfin = P.flags & FLAG_FINAL;
    if (wantfin) { R->fin.assign((size_t)n * FIN, 0.0); for (int64_t i = 0; i < n; i++) { double* f = &R->fin[i * FIN]; memcpy(f, sp + i * 3, 24); memcpy(f + 3, rays + i * 10, 80); f[18] = (double)(i + prm->ray_id_base + P.base); f[21] = 3; } }
    std::vector<RayState> cur(n); std::vector<int64_t> slot(n);
    for (int64_t i = 0; i < n; i++) {
        RayState& q = cur[i]; q.p = ld3(sp + i * 3);
        const double* r = rays + i * 10;
        V3 d = ld3(r), pl = ld3(r + 3);
        q.r.d = (1.0 / sqrt(dot(d, d))) * d; q.r.pol = (1.0 / sqrt(dot(pl, pl))) * pl; // RayTracer2.py:221-222
        q.r.s0 = r[6]; q.r.s1 = r[7]; q.r.s2 = r[8]; q.r.s3 = r[9];
        q.ray_id = i + prm->ray_id_base; q.parent = i; q.hash = 0; q.k = 0; q.smix_last = 1; q.six_last = 0; q.flags = RF_PRIMARY;
    }
    int step = 0;
    while (!cur.empty() && step < P.K) {
        int k = step++;
        size_t nl = cur.size(); R->live[k] = (int64_t)nl;
        std::vector<Bounce> B(nl);
        int64_t nsc = 0;
        const bool lean = P.flags == FLAG_SINGLE && !getenv("HS_NO_LEAN"); // the level kernel's tally-only step (bounce_phys_single)
        for (size_t i = 0; i < nl; i++) {
            if (lean) {
                RayState& q = cur[i]; Hit h; bounce_hit(G, P, q, true, h);
                LeanBounce lb; bounce_phys_single(G, P, q, h, lb);
                Bounce& b = B[i]; b.scat = lb.scat; b.surf_sc = lb.surf_sc; b.smix_next = lb.smix_next; b.six_next = lb.six_next; b.fa = lb.fa; b.fb = 0;
                b.bulk_abs = lb.bulk_abs; b.surf_abs = lb.surf_abs; b.inc_intensity = lb.inc_intensity; b.p = lb.p; b.refr = lb.child; b.refl = lb.child; b.refl.s0 = 0;
            } else bounce(G, P, cur[i], B[i]);
            nsc += B[i].scat;
        }
        R->scat[k] = nsc;
        std::vector<RayState> nxt;
        if (nsc > 0) {
            for (size_t i = 0; i < nl; i++) {
                const Bounce& b = B[i];
                if (!b.scat) { tally(R->table, S, k, 2, cur[i].smix_last, P.base, b.inc_intensity); continue; }
                if (b.surf_sc) tally(R->table, S, k, 0, b.smix_next, P.base, b.surf_abs);
                tally(R->table, S, k, 1, b.smix_next, P.base, b.bulk_abs);
                tally(R->table, S, k, 3, b.smix_next, P.base, (b.fa ? 0 : b.refr.s0) + (b.fb ? 0 : b.refl.s0));
                tally(R->table, S, k, 4, b.smix_next, P.base, (b.fa ? b.refr.s0 : 0) + (b.fb ? b.refl.s0 : 0));
                if (P.flags & FLAG_FULL) { size_t o = R->rec[k].size(); R->rec[k].resize(o + REC); fill_record(cur[i], b, P, &R->rec[k][o], 1); }
                if (wantfin && (cur[i].flags & RF_PRIMARY)) fill_final(cur[i], b, P, &R->fin[(cur[i].ray_id - prm->ray_id_base) * FIN]);
            }
            for (size_t i = 0; i < nl; i++) if (B[i].scat && B[i].fa) nxt.push_back(child_of(cur[i], B[i], false, true));
            for (size_t i = 0; i < nl; i++) if (B[i].scat && B[i].fb) nxt.push_back(child_of(cur[i], B[i], true, B[i].fa != 0));
        }
        cur.swap(nxt);
    }
    R->num_steps = step; *out = R; return 0;
}
extern "C" int hs_num_steps(Result* r) { return r->num_steps; }
extern "C" int64_t hs_step_count(Result* r, int k) { return (int64_t)(r->rec[k].size() / REC); }
extern "C" void hs_step_copy(Result* r, int k, double* dst) { if (!r->rec[k].empty()) memcpy(dst, r->rec[k].data(), r->rec[k].size() * 8); }
extern "C" void hs_table(Result* r, double* dst) { memcpy(dst, r->table.data(), (size_t)r->num_steps * 5 * r->S * 2 * 8); }
extern "C" void hs_counts(Result* r, int64_t* live, int64_t* scat) { memcpy(live, r->live.data(), r->K * 8); memcpy(scat, r->scat.data(), r->K * 8); }
extern "C" void hs_final(Result* r, double* dst) { if (!r->fin.empty()) memcpy(dst, r->fin.data(), r->fin.size() * 8); }
extern "C" void hs_free(Result* r) { delete r; }
// frame program of a geometry (test aid): copies up to cap records of PREC doubles, returns the record count
extern "C" int hs_program(const double* table, int S, const double* atoms, const uint32_t* code, int frames_on, double* dst, int cap) {
    std::vector<double> recs, prog; std::vector<uint32_t> smeta;
    build_device_tables(table, S, atoms, 0, code, recs, smeta);
    build_program(table, S, atoms, code, frames_on != 0, recs, smeta, prog);
    int n = (int)(prog.size() / PREC) - 1;
    for (int i = 0; i < n && i < cap; i++) memcpy(dst + (size_t)i * PREC, &prog[(size_t)i * PREC], PREC * 8);
    return n;
}
