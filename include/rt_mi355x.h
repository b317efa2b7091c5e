/*
 * rt_mi355x.h -- C ABI of librt_mi355x.so, the MI355X (gfx950) implementation of the
 * RayTracer2 propagation loop.
 *
 * The reference (cericdahl/RayTracer) has no FFI: its boundary is the Python function
 *   RayTracer2(ray_startingpoints, rays, surfacelist, max_scat, min_travel_len, follow_thresh,
 *              tir_handling, full_output, singlechild, output_raytable)
 * (python/RayTracer2.py:185, MATLAB/RayTracer2.m:177).  This header is what a ctypes/cffi binding of
 * that function calls; raytracer_amd/RayTracer2.py is such a binding and keeps the reference signature.
 *
 * Everything is plain C: pointers + sizes, int return codes (0 = ok, <0 = error, text from
 * rt_last_error()).  No exception crosses this boundary.  Host buffers are owned by the caller; device
 * buffers are owned by the context (or borrowed, see rt_trace_device).
 *
 * ---------------------------------------------------------------------------------------------
 * Data formats (shared with the CPU oracle in oracle/, which re-implements the readers on its own)
 * ---------------------------------------------------------------------------------------------
 * Surface table: n_surf rows of RT_ROW doubles, one row per `surface` object
 * (python/surface.py:51-62 + optional attributes python/RayTracer2.py:232-248):
 *   [0]      shape id        RT_SHAPE_*     (python/RayToShape.py:18-29)
 *   [1]      surface type id RT_STYPE_*     (python/RayTracer2.py:301-309; anything else = normal)
 *   [2..14]  geometry parameters, meaning depends on shape:
 *              plane       point(3) normal(3)            (RayToPlane.py:5, normal normalised :27-28)
 *              sphere      center(3) radius              (RayToSphere.py:18)
 *              cylinder    center(3) axis(3) radius      (RayToCylinder.py:53, axis normalised :73-74)
 *              quadsurface Q(9, row-major) P(3) R        (MATLAB/RayToX/RayToQuadsurface.m:55)
 *              torus       center(3) axis(3) r1 r2       (MATLAB/RayToX/RayToTorus.m:50, axis normalised :75-76)
 *   [15] n_outside  [16] n_inside  [17] absorption
 *   [18] abslength_outside [19] abslength_inside [20] rayleigh_outside [21] rayleigh_inside
 *   [22..26] unifiedparams[5] = {sigma_alpha, P_diffuse_reemit, P_lobe, P_spike, P_back}
 *              (python/UnifiedReflectorModel.py:95, MATLAB/Physics/UnifiedReflectorModel.m:48)
 *   [27] bounds program offset (index into code[]) [28] bounds program length  [29..31] reserved (0)
 *
 * Bounds program: the reference's `inbounds_function` is an opaque callable p(N,3,M) -> bool(N,M)
 * (python/surface.py:19-32).  The host compiles it to a postfix boolean program over "atoms"; the
 * program is evaluated per candidate hit point with a bit-stack (depth <= 32).
 *   code[i] = (op << 24) | arg ;  ops RT_BOP_*
 *   atom a  = atoms[a*RT_ATOM .. a*RT_ATOM+11] = {kind, cmp, rhs, c0..c8}
 *     kind RT_ATOM_QUAD : v = c0 x^2 + c1 y^2 + c2 z^2 + c3 xy + c4 xz + c5 yz + c6 x + c7 y + c8 z
 *     kind RT_ATOM_LIN  : v = c6 x + c7 y + c8 z
 *     kind RT_ATOM_COORD: v = p[(int)c0]                                     (bit-exact `p[:,k,:] < c`)
 *     kind RT_ATOM_ATAN2: v = atan2(c0 x + c1 y + c2 z + c3, c4 x + c5 y + c6 z + c7)
 *     result = v <cmp> rhs, cmp in RT_CMP_* ; comparisons with NaN are false (NE: true), as in numpy.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ROW 32
#define RT_ATOM 12
#define RT_MAX_SURF 256

enum { RT_SHAPE_PLANE = 0, RT_SHAPE_SPHERE = 1, RT_SHAPE_CYLINDER = 2, RT_SHAPE_QUADSURFACE = 3, RT_SHAPE_TORUS = 4 };
enum { RT_STYPE_NORMAL = 0, RT_STYPE_DIFFUSE = 1, RT_STYPE_UNIFIED = 2, RT_STYPE_RETRO = 3 };
enum { RT_BOP_ATOM = 1, RT_BOP_AND = 2, RT_BOP_OR = 3, RT_BOP_NOT = 4, RT_BOP_TRUE = 5, RT_BOP_FALSE = 6, RT_BOP_XOR = 7 };
enum { RT_ATOM_QUAD = 0, RT_ATOM_LIN = 1, RT_ATOM_ATAN2 = 2, RT_ATOM_COORD = 3 };
enum { RT_CMP_LT = 0, RT_CMP_LE = 1, RT_CMP_GT = 2, RT_CMP_GE = 3, RT_CMP_EQ = 4, RT_CMP_NE = 5 };

/* row offsets */
enum {
    RT_R_SHAPE = 0, RT_R_STYPE = 1, RT_R_GEOM = 2, RT_R_NOUT = 15, RT_R_NIN = 16, RT_R_ABS = 17,
    RT_R_ABSL_OUT = 18, RT_R_ABSL_IN = 19, RT_R_RAYL_OUT = 20, RT_R_RAYL_IN = 21, RT_R_UNIFIED = 22,
    RT_R_BOFF = 27, RT_R_BLEN = 28
};

/* compat bits: which conventions/quirks of the numpy port are reproduced (SURVEY.md quirk ledger) */
enum {
    RT_COMPAT_PY_INDEX = 1, /* Q1/Q2/Q4/Q5: 0-based `n*orientation` indices; surface 0 loses its sign and is
                               tallied on both orientations (python/RayTracer2.py:371,475-476); default */
    RT_COMPAT_PY_ROOT2 = 2  /* Q16: second sphere/cylinder root = -(first) (python/RayToSphere.py:47) */
};

/* flags */
enum {
    RT_FLAG_FULL_OUTPUT = 1, /* keep one record per interaction (python/RayTracer2.py:489-507) */
    RT_FLAG_SINGLECHILD = 2, /* dice between reflected/refracted child (python/RayTracer2.py:458-467) */
    RT_FLAG_FINAL_ROWS = 4   /* keep the per-ray final row (see RT_FINAL_COLS) */
};

typedef struct rt_params {
    int32_t max_scatters;    /* K      (RayTracer2.py:209-212) */
    int32_t flags;           /* RT_FLAG_* */
    int32_t compat;          /* RT_COMPAT_* */
    int32_t unified_maxiter; /* cap of the UNIFIED rejection loops (reference: unbounded); 0 = default 10000 */
    double min_travel_length; /* (RayTracer2.py:203-206) */
    double follow_threshold[2]; /* refracted, reflected (RayTracer2.py:194-200, 518-519) */
    double tir_handling;     /* <0: tangent ray with zero intensity; >=0: copy of reflected * value
                                (RefractionReflectionAtInterface.m:217-225) */
    uint64_t seed;           /* Philox key */
    int64_t ray_id_base;     /* global id of ray 0 of this call (sharding over GPUs / chunks) */
    int64_t ray_id_stride;   /* id of ray i = ray_id_base + i * stride; 0 = 1.  G for a rank that takes every G-th ray of a
                                spatially ordered source (camera pixels, GenerateRaysFromCamera.m:44-46), so that in
                                ray-splitting mode every rank gets the same mix of long and short lineages */
} rt_params;

/* One record per interaction = one row of the reference's rayInterfaces arrays (python/rayInterfaces.py:4-20),
 * RT_REC_COLS doubles:
 *  [0..9] incoming_ray [10..19] refracted_ray [20..29] reflected_ray [30..32] intersection_point
 *  [33..35] surface_normal (mechanical) [36] ray_index [37] surface_index [38] distance_traveled
 *  [39] n_incident [40] n_transmitted [41] bulkabs_incident [42] bulkabs_transmitted
 *  [43] rayleigh_incident [44] rayleigh_transmitted
 *  extras (not in the reference): [45] orientation (+1 into / -1 out of the surface) [46] 0-based surface number
 *  [47] 1 if Rayleigh scatter else 0 */
#define RT_REC_COLS 48

/* Final row per initial ray (the reduction python/BubbleImageCreator.py:149-161 + finalRays.py makes from
 * ray_interfaces: last record of each ray id), RT_FINAL_COLS doubles:
 *  [0..2] position of last interaction (start point if none) [3..5] direction [6..8] polref [9..12] Stokes s0..s3
 *  of the ray arriving there, [13..15] surface normal, [16] surface_index (reference convention: number * orientation) [17] orientation
 *  [18] ray_index [19] number of interactions [20] s0 still followed after it
 *  [21] status: 0 escaped geometry, 1 absorbed / below threshold, 2 max_scatters reached, 3 none (never hit)
 *  [22..24] direction, [25..27] polref, [28..30] s1..s3 of the ray followed after it (refracted child if followed,
 *  else reflected) [31] reserved.  Non-dice mode: rows describe the lineage that keeps following the refracted
 *  child when there is one.  A ray that never interacts keeps its input values in [0..12] and status 3. */
#define RT_FINAL_COLS 32

typedef struct rt_ctx rt_ctx;

/* Create a context on HIP device `device`.  Replaces nothing in the reference (it has no device). */
int rt_init(int device, rt_ctx** out);
void rt_free(rt_ctx* ctx);
const char* rt_last_error(rt_ctx* ctx); /* ctx may be NULL: the calling thread's last rt_init error */

/* Upload the geometry.  Replaces the per-call walk over `surfacelist` objects (RayTracer2.py:234-248, 288-374). */
int rt_set_surfaces(rt_ctx* ctx, const double* table, int32_t n_surf,
                    const double* atoms, int32_t n_atoms, const uint32_t* code, int32_t n_code);

/* Upload the Rayleigh azimuth inverse-CDF table (MATLAB/Physics/RayleighScattering.m:55-60), n = precision+1. */
int rt_set_rayleigh_lut(rt_ctx* ctx, const double* lut, int32_t n);

/* Trace n rays given in the reference's host layout: start[n][3], rays[n][10] row-major
 * (RayTracer2.py:185 arguments 1-2).  Blocking.  Results are fetched with rt_result_*. */
int rt_trace(rt_ctx* ctx, const double* start, const double* rays, int64_t n, const rt_params* prm);

/* Same, rays already resident in HBM as structure-of-arrays: d_soa[f*n + i], f = 0..2 start point,
 * 3..12 the ten ray columns.  The pointer is a device pointer (e.g. torch tensor.data_ptr()). */
int rt_trace_device(rt_ctx* ctx, const double* d_soa, int64_t n, const rt_params* prm);

/* rt_trace_device for callers that cannot allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc, code that has set up a HIP graph or a pool and must not see it move.  The call itself is BLOCKING and runs
 * on the context's own stream (it ends with a stream synchronisation and cannot be recorded into a graph capture).
 * NO COUNTERPART IN THE REFERENCE (it has no device and no allocator to avoid).  Tally-only
 * single-child runs (prm->flags == RT_FLAG_SINGLECHILD, thresholds >= 0), at most RT_NOALLOC_MAX_RAYS rays and
 * RT_NOALLOC_MAX_SCATTERS steps per call; more is refused with an error.  Every device buffer beyond the caller's
 * inputs and outputs (ray queues, nearest hits, ray lists, counters) lives in a static device array of
 * RT_NOALLOC_SCRATCH_BYTES that each GPU the library runs on holds until the process ends; the call itself makes no
 * hipMalloc / hipFree and touches no pooled buffer.  Outputs go to the caller's device memory: d_table
 * [max_scatters][5][n_surf][2] (rows of steps in which nothing scattered are zero, as rt_result_table returns them;
 * the reference's num_scatters = 1 + the last step with n_live > 0) and d_step_counts [2][max_scatters] = n_live,
 * n_scat of rt_result_step_counts.  Blocking.  All calls on one GPU share that GPU's array: calls that overlap on one
 * GPU (two contexts, two threads) race - serialise them. */
#define RT_NOALLOC_MAX_RAYS 262144
#define RT_NOALLOC_MAX_SCATTERS 4094
#define RT_NOALLOC_SCRATCH_BYTES (80u << 20)
int rt_trace_device_noalloc(rt_ctx* ctx, const double* d_soa, int64_t n, const rt_params* prm,
                            double* d_table /* device */, uint64_t* d_step_counts /* device */);

/* Generate n isotropic unpolarised unit-intensity rays from one point directly in HBM (the source of
 * MATLAB/Examples/RunExamples/RunSBCGeometry.m:45-68) and trace them; no host arrays involved. */
int rt_trace_point_source(rt_ctx* ctx, const double origin[3], int64_t n, const rt_params* prm);

/* The rays rt_trace_point_source would trace, copied to the host in the reference layout (for parity tests). */
int rt_generate_point_source(rt_ctx* ctx, const double origin[3], int64_t n, uint64_t seed, int64_t ray_id_base,
                             double* start /*[n][3]*/, double* rays /*[n][10]*/);

/* Camera of MATLAB/Examples/ImagingExamples/GenerateRaysFromCamera.m:6-7 (python/GenerateRaysFromCamera.py is not
 * runnable) plus what the geometry builders append to every direction (createGeometry.py:1113-1122,
 * CreateArBCGeometry.m:632-637): starting point, polarisation reference, Stokes vector. */
enum { RT_LENS_TAN = 0, RT_LENS_THETA = 1, RT_LENS_SIN = 2 };   /* lens_type 'tan' (default) | 'theta' | 'sin', :58-67 */
typedef struct rt_camera {
    int32_t resolution[2];   /* pixels along i and j */
    int32_t lens_type;       /* RT_LENS_* */
    int32_t n_distortion;    /* number of radial distortion coefficients used (<= 8), :54-56 */
    double pixel_pitch[2], pixel_center[2], focal_length, pitch, yaw, roll, distortion[8];
    double position[3], polref[3], stokes[4];
} rt_camera;

/* One ray per pixel, generated in HBM, traced.  Ray id = ray_id_base + (j-1)*resolution[0] + (i-1), the row order of
 * GenerateRaysFromCamera.m:44-46.  Replaces GenerateRaysFromCamera + the host arrays of BubbleImageCreator.py:129. */
int rt_trace_camera(rt_ctx* ctx, const rt_camera* cam, const rt_params* prm);
/* The same for n pixels first, first + stride, ... (stride = prm->ray_id_stride; ray i has id ray_id_base + i * stride): a
 * rank's share of the sensor, contiguous or interleaved, without host arrays (no reference counterpart). */
int rt_trace_camera_range(rt_ctx* ctx, const rt_camera* cam, int64_t first, int64_t n, const rt_params* prm);
/* Rays of pixels [first, first+n) copied to the host in the reference layout, pixels[n][2] = (i, j) 1-based:
 * the return values of GenerateRaysFromCamera.m:6 with the constant columns appended. */
int rt_generate_camera_rays(rt_ctx* ctx, const rt_camera* cam, int64_t first, int64_t n,
                            double* start /*[n][3]*/, double* rays /*[n][10]*/, int32_t* pixels /*[n][2]*/);

/* Tuning knobs (no reference counterpart): "workspace_gb" (device memory for records/queues, default 32),
 * "level_steps" / "level_min_rays" (runs are traced step by step for at most that many steps and while
 * at least that many rays are live, the persistent kernel takes the rest; no cap / by run mode: 786432 for single-child
 * runs that keep tallies only or records, 524288 for those that keep final rows, 0 for ray-splitting runs - a negative
 * value restores these), "defer_unified" (0: rays ending
 * on UNIFIED surfaces are not given a physics launch of their own), "hit_split" / "hit_split_lanes" (blocks of the nearest-hit
 * launch per queue segment: alone on the GPU, default 128 = one per tile of 256 rays; beside another lane's physics launches, default 12), "seg_per_cu" (queue segments per CU; default
 * 2 for single-child runs up to 16M rays, else 8), "sync_every" / "sync_rays_max" (tally-only single-child runs below that
 * many live rays launch up to that many steps between two host synchronisations; 8 / 16M), "lanes" / "lanes_min_rays" /
 * "lanes_alternate" (tally-only runs - dice or ray splitting - of at least that many rays are traced as that many parts (1..4) at the same time, each on
 * a stream and host thread of its own by a context that shares the geometry, taking turns with the nearest-hit launch so
 * that it runs beside the other lane's physics launches: +10 % at 33.5M rays; per-kernel timings (rt_result_timing*) are
 * then sums over launches that overlap in time; default 2 / 2M / 1), "frames" (0: the
 * reference's formulas for every surface instead of the frame program), "refill_min" (idle lanes per wave that
 * trigger a refill of the persistent kernel, default 4), "blocks_per_cu" (0 = occupancy query), "nodice_chunk" (rays
 * per pass in ray-splitting mode, default 2^20), "lds_budget_kb", "match_max_level" (rt_match_endpoints: deepest level of
 * the search tree, default 10), "stream_overlap" (0: record streaming copies and calls the sink between the kernels of
 * two chunks, as up to r1k), "stream_stage_gb" (bytes of records per chunk handed to a record sink, default 2; two
 * pinned host buffers of that size are kept), "upload_chunk" (rays per piece in which host inputs
 * are staged through two pinned buffers, default 2^21; 0: one hipMemcpy from the caller's pageable arrays),
 * "accumulate_stage_gb" (the same for records that accumulate for
 * rt_result_records in single-child runs, default 0.25), "release_workspace" (free the pooled device buffers and those two). */
int rt_set_option(rt_ctx* ctx, const char* key, double value);

/* Record streaming (no reference counterpart: the reference keeps every rayInterfaces array in
 * memory, python/RayTracer2.py:489-507).  With a sink set, a full_output trace hands over the records of each chunk of
 * rays as soon as that chunk is traced instead of accumulating them: n records in device order with the same ordering
 * keys as rt_result_records (parent = index within this call of the sink, or -1), for the rays
 * [first_ray_id, first_ray_id + n_rays) - in single-child mode all interactions of those rays.  The pointers are valid
 * during the call only.  The records of a chunk are packed on the device and copied into one of two pinned host
 * stages on a second stream; the sink is called on rt_trace's thread for chunk c-1 while chunk c is being copied, and
 * chunk c+1 is traced under the same copy.  Chunk size: "stream_stage_gb" worth of records (single-child: that many
 * bytes / (408 * max_scat) rays) / "nodice_chunk" (ray splitting).  A non-zero return stops the trace (rt_trace
 * returns -4).  sink = NULL restores accumulation. */
typedef int (*rt_record_sink)(void* user, int64_t n, const double* rec /*[n][RT_REC_COLS]*/, const int32_t* step, const int64_t* ray_id,
                              const int64_t* parent, const int32_t* flags, int64_t first_ray_id, int64_t n_rays);
int rt_set_record_sink(rt_ctx* ctx, rt_record_sink sink, void* user);

/* Image formation after the two traces of BubbleImageCreator: every LED ray that ends on one of `surfaces` gives a
 * Gaussian-weighted share of its intensity to the camera ray ending nearest to it on the same surface.  Replaces the
 * O(N_led * N_cam) loop of python/BubbleImageCreator.py:278-357 (MATLAB kept there as comments; the numpy port stops
 * at :250) with a uniform-grid search on the device; the sequential rule of :302-311 is reproduced exactly, the
 * "distance == 0 means unset" sentinel included.  cam_rows / led_rows are final rows (RT_FINAL_COLS doubles each, see
 * above) of the camera and LED traces; `surfaces` lists the light-giving surfaces in the convention of column [16]
 * (sign ignored, as :287-288 compare both signs).  Rows with no interaction, with a non-finite end point or on other
 * surfaces take no part.  Outputs: pixel_intensity[n_cam] (the sum of :343-349, accumulated with atomics: order of
 * addition differs from run to run at the 1e-16 level); optional per LED ray: led_cam = matched camera row or -1,
 * led_dist = its distance, led_add = the intensity it added (int_addition of :338; summing led_add by led_cam in LED
 * order reproduces the reference's order of addition).  With RT_MATCH_DEVICE_POINTERS every array argument except
 * `surfaces` is a device pointer (e.g. torch tensors) and nothing crosses PCIe. */
typedef struct rt_match_params {
    double max_dist;          /* 5              :321  pairs further apart give nothing; also the grid cell size */
    double angle_den;         /* 2*10*pi/180    :338  denominator under (angle between the two rays)^2 */
    double cam_incidence_den; /* 2*45*pi/180    :338  under (camera ray's incident angle)^2 */
    double led_incidence_den; /* 2*45*pi/180    :338  under (LED ray's incident angle)^2 */
    double dist_den;          /* 2*2^2          :338  under distance^2 */
    int32_t flags;            /* RT_MATCH_* */
    int32_t reserved;
} rt_match_params;
enum { RT_MATCH_DEVICE_POINTERS = 1 };
int rt_match_endpoints(rt_ctx* ctx, const double* cam_rows /*[n_cam][RT_FINAL_COLS]*/, int64_t n_cam, const double* led_rows /*[n_led][RT_FINAL_COLS]*/,
                       int64_t n_led, const int32_t* surfaces, int32_t n_surfaces, const rt_match_params* mp,
                       double* pixel_intensity /*[n_cam]*/, int64_t* led_cam /*[n_led] or NULL*/, double* led_dist /*[n_led] or NULL*/,
                       double* led_add /*[n_led] or NULL*/);

/* Results of the last trace. */
int rt_result_num_steps(rt_ctx* ctx, int32_t* num_steps);           /* rows of absorption_table (RayTracer2.py:542) */
/* Ray-splitting runs (no RT_FLAG_SINGLECHILD) are traced in passes of `nodice_chunk` rays, and a pass cannot know in
 * advance how many rays and records it will hold at once (the reference grows its arrays as it goes,
 * python/RayTracer2.py:531-540).  A pass whose queue or record buffer overflows is traced again from its first ray in
 * passes a quarter the size - its tallies went into a table of its own, which is dropped - and the call goes on; only when
 * that has failed eight times (or the pass is down to 256 rays) does rt_trace return -3 (the caller then lowers
 * nodice_chunk and starts over).  *n: how many passes of the last trace were repeated that way (diagnostic). */
int rt_result_pass_retries(rt_ctx* ctx, int32_t* n);
int rt_result_table(rt_ctx* ctx, double* table /* [num_steps][5][n_surf][2] */);
int rt_result_step_counts(rt_ctx* ctx, int64_t* n_live /*[K]*/, int64_t* n_scat /*[K]*/);
/* The table of the last trace without the trip through host memory: all max_scatters rows [K][5][n_surf][2] (rows of steps
 * in which nothing scattered, and rows behind num_steps, are zero) copied into the caller's device buffer, and optionally
 * the counters [2][K] = n_live, n_scat.  For the one collective of a multi-GPU run: the all-reduce of the tallies takes
 * the device buffer as it is (raytracer_amd/multigpu.py, bench.py --gpus N).  No reference counterpart. */
int rt_result_table_device(rt_ctx* ctx, double* d_table /* device */, uint64_t* d_step_counts /* device, or NULL */);
int rt_result_num_records(rt_ctx* ctx, int64_t* n);
/* records in device order + ordering metadata; the host wrapper sorts them into the reference's order */
int rt_result_records(rt_ctx* ctx, double* rec /*[n][RT_REC_COLS]*/, int32_t* step /*[n]*/,
                      int64_t* ray_id /*[n]*/, int64_t* parent /*[n]*/, int32_t* flags /*[n]*/);
/* The arrays of python/rayInterfaces.py:4-20 for one step without a copy of the whole record table: records
 * order[0..n) (indices into the record table; NULL = the first n) split into fields, field f = record columns
 * [col0[f], col0[f] + ncols[f]) as a dense [n][ncols[f]] array at outs[f].  Any of rec, step, ... of
 * rt_result_records may be NULL, so the wrapper fetches the ordering metadata alone and then calls this per step. */
int rt_result_record_fields(rt_ctx* ctx, const int64_t* order, int64_t n, int32_t nfields, const int32_t* col0, const int32_t* ncols,
                            double* const* outs);
/* The same for any table of record rows, e.g. the one a record sink is handed (no context: plain host work).
 * Returns 0, -1 (bad column range) or -2 (record index out of range). */
int rt_split_record_fields(const double* rows /*[n_rows][RT_REC_COLS]*/, int64_t n_rows, const int64_t* order, int64_t n, int32_t nfields,
                           const int32_t* col0, const int32_t* ncols, double* const* outs);
/* Host work of the drop-in call, off the interpreter (plain host code, no context).
 * rt_normalize_rays: the reference normalises the caller's `rays` in place (python/RayTracer2.py:221-222: columns 0-2 and
 * 3-5, each divided by |sqrt(sum of squares)|); the same operations in the same order, on several host threads.
 * rt_order_records: the reference's row order (python/RayTracer2.py:531-540: step 0 in input order, step k + 1 =
 * [refracted children in the order of their parents ; reflected children in the order of their parents]) from the ordering
 * metadata of rt_result_records: order[0..n) = record indices, step after step; counts[k] = records of step k.  Returns 0,
 * -1 (bad arguments) or -2 (metadata that is not a tree of that form, e.g. ray ids too sparse for a table: sort instead). */
int rt_normalize_rays(double* rays /*[n][10]*/, int64_t n, double* max_s0 /* NULL, or: the largest rays[:,6] (NaN if any is), RayTracer2.py:252 */);
int rt_order_records(const int32_t* step, const int64_t* ray_id, const int64_t* parent, const int32_t* flags, int64_t n, int32_t num_steps,
                     int64_t* order /*[n]*/, int64_t* counts /*[num_steps]*/);
int rt_result_final(rt_ctx* ctx, double* rows /*[n_rays][RT_FINAL_COLS]*/);
/* Final rows without the trip through host memory (no reference counterpart; the reference's reduction,
 * python/BubbleImageCreator.py:134-165, lives in host arrays): traces with RT_FLAG_FINAL_ROWS write row (ray id -
 * ray_id_base) straight into d_rows, a device buffer of the caller's with room for n_rows rows (e.g. a torch tensor),
 * and rt_result_final has nothing to return.  rt_match_endpoints takes such buffers as they are.  d_rows = NULL
 * restores the host copy. */
int rt_set_final_rows_device(rt_ctx* ctx, double* d_rows, int64_t n_rows);
/* kernel time of the last trace (or rt_match_endpoints call) measured with HIP events on the context's stream, and launch count */
int rt_result_timing(rt_ctx* ctx, double* kernel_ms, int32_t* n_launches);
/* the same split by kernel: [0] k_trace (persistent bounce kernel), [1] k_level_hit, [2] k_level_phys */
int rt_result_timing_by_kernel(rt_ctx* ctx, double ms[3], int32_t launches[3]);

/* device properties the host side needs (CU count etc.) */
int rt_device_info(rt_ctx* ctx, int32_t* n_cu, int64_t* hbm_bytes, char* name, int32_t name_len);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
