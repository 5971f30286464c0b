// api.hip -- C-ABI entry points (include/igs_rast.h) and host-side orchestration on a HIP stream.
// Counterpart of CudaRasterizer::Rasterizer::{forward,backward,markVisible}
// (DGR/cuda_rasterizer/rasterizer_impl.cu:176-188, 254-425, 429-571).
#include <atomic>
#include "common.h"
#include <math.h>
#include "../../include/igs_rast.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>

static thread_local char g_err[512] = "";
static int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    if (e != hipSuccess) snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    else snprintf(g_err, sizeof g_err, "%s", what);
    return code;
}
#define HIP_TRY(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(IGS_RAST_E_HIP, what, e_); } while (0)
// IGS_TRACE_LAUNCHES=1 (debugging aid): synchronise after every launch like debug mode and name it on stderr once it has completed --
// after a GPU memory fault (which aborts the process) the launch that follows the last name printed is the one that faulted
static const bool g_trace_launches = getenv("IGS_TRACE_LAUNCHES") != nullptr;
#define DBG_SYNC(what) do { if (debug || g_trace_launches) { hipError_t e_ = hipStreamSynchronize(s); if (e_ != hipSuccess) return fail(IGS_RAST_E_HIP, what, e_); \
                                                              if (g_trace_launches) { fprintf(stderr, "[igs] completed: %s\n", what); fflush(stderr); } } } while (0)

// pinned status slot + event: one per host thread AND device, kept until the thread ends (a blend kernel still running on the
// device used before may yet post into its slot, so switching devices never frees one)
struct HostSlot { uint32_t* pinned = nullptr; uint32_t* pinned_dev = nullptr; hipEvent_t ev = nullptr; };
#define IGS_MAX_DEVICES 64
struct HostSlots {
    HostSlot slot[IGS_MAX_DEVICES];
    ~HostSlots() { for (auto& h : slot) if (h.pinned) { (void)hipHostFree(h.pinned); (void)hipEventDestroy(h.ev); } }
};
static thread_local HostSlots g_slots;
static thread_local HostSlot g_slot;                // the current device's slot (a copy of the table entry)
static thread_local hipStream_t g_status_stream = nullptr;
static thread_local uint32_t g_host_seq = 0;      // sequence number of the last status the blend kernel was asked to post
static thread_local uint32_t g_nowait_seq = 0;    // ... and the one baked into the last igs_rast_forward_nowait (a capture): what igs_rast_last_status expects
static int ensure_slot()
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
    if (dev < 0 || dev >= IGS_MAX_DEVICES) return fail(IGS_RAST_E_INVALID, "device ordinal out of range");
    HostSlot& h = g_slots.slot[dev];
    if (!h.pinned) {
        HIP_TRY(hipHostMalloc((void**)&h.pinned, (COUNTER_SHARDS + 1) * COUNTER_SHARD_STRIDE * 4, hipHostMallocDefault), "hipHostMalloc");
        HIP_TRY(hipHostGetDevicePointer((void**)&h.pinned_dev, h.pinned, 0), "hipHostGetDevicePointer");
        HIP_TRY(hipEventCreateWithFlags(&h.ev, hipEventDisableTiming), "hipEventCreate");
    }
    g_slot = h;
    return 0;
}

static double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static double wait_limit_s()
{
    static double lim = -1.0;
    if (lim < 0.0) { const char* e = getenv("IGS_RAST_WAIT_TIMEOUT_S"); lim = e ? atof(e) : 10.0; if (!(lim > 0.0)) lim = 10.0; }
    return lim;
}

// Waits until the blend kernel of the current slab-binned frame has posted {R, overflow, prefilter flag} into pinned host
// memory: a poll on the sequence word instead of an event -- an event record between blend_fwd and blend_bwd costs the stream
// a ~6 us bubble per step, and the first workgroup posts at the START of the kernel, so the host is released earlier too.
// Bounded: a stream that reports an error, a stream that drains without the post, or IGS_RAST_WAIT_TIMEOUT_S (default 10)
// seconds of wall clock without it all end the wait with IGS_RAST_E_HIP.
static int wait_status(hipStream_t s)
{
    volatile uint32_t* seq = (volatile uint32_t*)&g_slot.pinned[3];
    double t0 = 0.0;
    for (long spins = 0;; spins++) {
        if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == g_host_seq) return 0;
        if ((spins & 0x3FFF) == 0x3FFF) {
            const hipError_t q = hipStreamQuery(s);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(IGS_RAST_E_HIP, "stream error while waiting for the frame status", q);
            if (q == hipSuccess && __atomic_load_n(seq, __ATOMIC_ACQUIRE) != g_host_seq)
                return fail(IGS_RAST_E_HIP, "the frame status was never posted");
            const double t = now_s();
            if (t0 == 0.0) t0 = t;
            else if (t - t0 > wait_limit_s())
                return fail(IGS_RAST_E_HIP, "timed out waiting for the frame status (IGS_RAST_WAIT_TIMEOUT_S)");
        }
    }
}

static int ceil_log2(uint32_t n) { int b = 0; while ((1u << b) < n) b++; return b; }

// ---- optional per-stage timing with HIP events on the caller's stream (roofline harness, bench.py) ----
enum { ST_PREPROCESS = 0, ST_DEPTH_SORT, ST_SCAN, ST_EMIT, ST_TILE_SORT, ST_RANGES, ST_BLEND_FWD, ST_MEMSET, ST_BLEND_BWD,
       ST_GEOM_BWD, ST_COUNT, ST_GAP = -1 };
struct Prof {
    bool on = false;
    int every = 1; long long frame = 0; bool active = false;     // marks are recorded on every `every`-th frame only
    static const int CAP = 1024;
    hipEvent_t ev[CAP]; int tag[CAP]; int n = 0; bool created = false;
    double ms[ST_COUNT] = {0}; long long cnt[ST_COUNT] = {0}; double r_sum = 0; long long calls = 0;
};
static Prof g_prof;
static void prof_collect()
{
    if (g_prof.n == 0) return;
    (void)hipEventSynchronize(g_prof.ev[g_prof.n - 1]);
    for (int i = 1; i < g_prof.n; i++) {
        const int t = g_prof.tag[i];
        if (t < 0) continue;                       // a mark that only starts a new interval
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[i - 1], g_prof.ev[i]) == hipSuccess) { g_prof.ms[t] += ms; g_prof.cnt[t]++; }
    }
    g_prof.n = 0;
}
// records "stage `tag` ended here" (tag = ST_GAP: only a start mark)
static void prof_mark(hipStream_t s, int tag)
{
    if (!g_prof.on || !g_prof.active) return;
    if (g_prof.n >= Prof::CAP - 1) {                // keep the last mark as the start of the next interval
        const hipEvent_t last = g_prof.ev[g_prof.n - 1];
        prof_collect();
        (void)last;
        (void)hipEventRecord(g_prof.ev[0], s); g_prof.tag[0] = ST_GAP; g_prof.n = 1;
        if (tag == ST_GAP) return;
    }
    (void)hipEventRecord(g_prof.ev[g_prof.n], s); g_prof.tag[g_prof.n] = tag; g_prof.n++;
}
// on = 0: off; on = N > 0: record stage marks on every N-th frame (an event record costs a few microseconds of stream time:
// marking every frame slows a 0.35 ms refine step by 10 %, marking every 8th by about 1 %)
extern "C" int igs_rast_profile_enable(int on)
{
    if (!on) prof_collect();
    // (the events are created here, not at the first mark: a thousand hipEventCreate calls take over a millisecond)
    if (on && !g_prof.created) { for (int i = 0; i < Prof::CAP; i++) (void)hipEventCreate(&g_prof.ev[i]); g_prof.created = true; }
    g_prof.on = on != 0; g_prof.every = on > 0 ? on : 1; g_prof.frame = 0; g_prof.active = g_prof.on;
    return 0;
}
static void prof_new_frame() { if (g_prof.on) { g_prof.active = (g_prof.frame % g_prof.every) == 0; g_prof.frame++; } }
extern "C" int igs_rast_profile_read(double* ms_sum, long long* count, double* r_sum, long long* calls, int reset)
{
    prof_collect();
    for (int i = 0; i < ST_COUNT; i++) { if (ms_sum) ms_sum[i] = g_prof.ms[i]; if (count) count[i] = g_prof.cnt[i]; }
    if (r_sum) *r_sum = g_prof.r_sum;
    if (calls) *calls = g_prof.calls;
    if (reset) { for (int i = 0; i < ST_COUNT; i++) { g_prof.ms[i] = 0; g_prof.cnt[i] = 0; } g_prof.r_sum = 0; g_prof.calls = 0; }
    return ST_COUNT;
}

extern "C" int igs_rast_version(void) { return IGS_RAST_VERSION; }
extern "C" const char* igs_rast_last_error(void) { return g_err; }
// workspace = 256-byte aligned { gacc[P][32] floats | 64 loss shards one cache line apart (refine step) }
static inline size_t ws_gacc_bytes(int P) { return ((size_t)(P > 0 ? P : 0) * GACC_F * 4 + 255) & ~(size_t)255; }
#define WS_LOSS_BYTES 4096
extern "C" size_t igs_rast_backward_workspace_bytes(int P) { return ws_gacc_bytes(P) + WS_LOSS_BYTES + 512; }

// what the previous forward on this host thread saw: sizes the binning buffer before R is known
struct BinHint { uint32_t slab = 0; };              // instance slots per tile used by the last frames (0 = default)
#define SLAB_MAX_BYTES (8ull << 30)                 // beyond this much slab scratch the global-sort path is used
static thread_local BinHint g_hint;
// igs_rast_forward_async leaves its host-side check of R to igs_rast_forward_finish
struct PendingFwd { bool active = false; uint32_t slab = 0; };
static thread_local PendingFwd g_pending;
// what the extended entry points ask of the forward on top of the reference's argument list
struct FwdExtra {
    bool defer_status = false;          // igs_rast_forward_async / igs_refine_step: do not wait for {R, overflow}
    bool no_latch = false;              // igs_rast_forward_nowait: ... and do not expect an igs_rast_forward_finish either
    bool raw_activations = false;       // igs_refine_step: opacities / scales / rotations are the raw optimiser leaves
    float* zero_gacc = nullptr;         // ... backward accumulators / loss shards the preprocess kernel zero-fills on the side
    float* zero_loss = nullptr;
    float* zero_loss2 = nullptr;
    bool scratch_clean = false;         // igs_refine_step_args::scratch_clean: the image buffer's binning counters are known clean
    bool skip_bwd_state = false;        // ... the loss is colour-only: blend_fwd need not store the geometry branches' backward state
    uint32_t plane_tag = 0;             // ... nonzero: a plane / depth / normal gradient will come back: keep Sigma^-1 per Gaussian under this tag
    // igs_refine_step with the L1 loss: run the colour-only blend backward inside the forward's tile kernel (blend_step.hip).  The
    // caller fills everything of the backward's arguments that forward_impl does not know (bg, gacc, l1_*, want_absgrad); the
    // list / record / geometry fields are set here.  *fused_ran reports whether that kernel was used (slab binning only).
    BlendBwdArgs* fused_bwd = nullptr;
    bool* fused_ran = nullptr;
    int* fused_instance = nullptr;
};
// where the last slab-binned forward left its device-side validity words (refine step guards)
struct LastFwd { const uint32_t* overflow = nullptr; const uint32_t* prefilter = nullptr; };
static thread_local LastFwd g_last_fwd;
// a slab-binned forward of this thread got as far as its binning kernel but not to its tile sort: some buffer's counters are not clean
static thread_local bool g_counters_dirty = false;
// one-shot promise for the NEXT forward of this thread (igs_rast_hint_scratch_clean)
static thread_local bool g_hint_clean = false;
extern "C" void igs_rast_hint_scratch_clean(int on) { g_hint_clean = on != 0; }

static int forward_impl(
    void* stream,
    igs_rast_alloc_fn geometry_buffer, void* geometry_user, igs_rast_alloc_fn binning_buffer, void* binning_user,
    igs_rast_alloc_fn image_buffer, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
    float* out_color, float* out_coord, float* out_mcoord, float* out_depth, float* out_mdepth, float* out_alpha,
    float* out_normal, int* radii, int require_coord, int require_depth, int debug,
    bool force_radix, uint64_t min_capacity, const FwdExtra& ex)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || width <= 0 || height <= 0) return fail(IGS_RAST_E_INVALID, "igs_rast_forward: bad sizes");
    if (P == 0) return 0;                                       // rasterize_points.cu:90: nothing is launched
    if (!geometry_buffer || !binning_buffer || !image_buffer) return fail(IGS_RAST_E_INVALID, "igs_rast_forward: NULL scratch callback");
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos || !background || !radii)
        return fail(IGS_RAST_E_INVALID, "igs_rast_forward: NULL required input");
    if (!out_color || !out_coord || !out_mcoord || !out_depth || !out_mdepth || !out_alpha || !out_normal)
        return fail(IGS_RAST_E_INVALID, "igs_rast_forward: NULL output");
    if (!colors_precomp && !shs) return fail(IGS_RAST_E_INVALID, "igs_rast_forward: neither shs nor colors_precomp");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(IGS_RAST_E_INVALID, "igs_rast_forward: neither scales/rotations nor cov3D_precomp");
    if (!colors_precomp && (M < (D + 1) * (D + 1) || D < 0 || D > 3)) return fail(IGS_RAST_E_INVALID, "igs_rast_forward: SH degree / coefficient count mismatch");
    if (int rc = ensure_slot()) return rc;

    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    const size_t Tn = (size_t)gx * gy, HW = (size_t)width * height;

    const GeomLayout GL(P);
    char* gbase = geometry_buffer(geometry_user, GL.total);
    if (!gbase) return fail(IGS_RAST_E_ALLOC, "geometry buffer callback returned NULL");
    gbase = align_ptr(gbase);
    const ImgLayout IL(HW, Tn);
    char* ibase = image_buffer(image_user, IL.total);
    if (!ibase) return fail(IGS_RAST_E_ALLOC, "image buffer callback returned NULL");
    ibase = align_ptr(ibase);

    float* rec = (float*)(gbase + GL.rec);
    uint32_t* tiles = (uint32_t*)(gbase + GL.tiles);
    uint32_t* keys_a = (uint32_t*)(gbase + GL.keys_a); uint32_t* keys_b = (uint32_t*)(gbase + GL.keys_b);
    uint32_t* vals_a = (uint32_t*)(gbase + GL.vals_a); uint32_t* vals_b = (uint32_t*)(gbase + GL.vals_b);
    uint32_t* ghist = (uint32_t*)(gbase + GL.hist);
    uint32_t* blocksum = (uint32_t*)(gbase + GL.blocksum);
    uint32_t* counters = (uint32_t*)(gbase + GL.counters);

    FwdParams fp;
    fp.P = P; fp.D = D; fp.M = M; fp.W = width; fp.H = height; fp.gx = gx; fp.gy = gy;
    fp.means3D = means3D; fp.shs = shs; fp.colors_precomp = colors_precomp; fp.opacities = opacities;
    fp.scales = scales; fp.rotations = rotations; fp.cov3D_precomp = cov3D_precomp;
    fp.scale_modifier = scale_modifier; fp.tan_fovx = tan_fovx; fp.tan_fovy = tan_fovy;
    fp.fy = height / (2.0f * tan_fovy); fp.fx = width / (2.0f * tan_fovx);       // rasterizer_impl.cu:288-289
    fp.kernel_size = kernel_size; fp.prefiltered = prefiltered;
    fp.view = viewmatrix; fp.proj = projmatrix; fp.campos = cam_pos;
    fp.raw_activations = ex.raw_activations ? 1 : 0;
    fp.zero_gacc = ex.zero_gacc; fp.zero_loss = ex.zero_loss; fp.zero_loss2 = ex.zero_loss2;
    if (ex.plane_tag) { fp.plane_cache = (float*)(gbase + GL.planes); fp.plane_tag = ex.plane_tag; }
    fp.zero_gacc_stride = ex.skip_bwd_state ? GACC_COMPACT_F : GACC_F;      // (colour-only loss <=> compact accumulator rows)
    g_last_fwd = LastFwd();

    const size_t counter_bytes = (COUNTER_SHARDS + 1) * COUNTER_SHARD_STRIDE * 4;
    uint32_t* ranges = (uint32_t*)(ibase + IL.ranges);
    uint32_t* point_list = nullptr;
    uint32_t R = 0;
    bool slab_pending = false;            // slab path: the host has not looked at R yet
    bool use_step_order = false; uint32_t* step_cursors = nullptr;      // fused refine step: see the tile sort below
    uint32_t slab_size = 0;                // slab size of this call
    const uint32_t* slab_stats = nullptr;

    if (!force_radix) {
        // ---------------- slab binning (default): nothing below needs the host to know R ----------------
        uint32_t* tile_count = (uint32_t*)(ibase + IL.tile_count);
        uint32_t* stats = (uint32_t*)(ibase + IL.stats);
        // slab size: sticky per thread, grown when a frame overflowed (the refine loop renders similar views over and over)
        uint64_t slab = g_hint.slab ? g_hint.slab : 1024;
        if (min_capacity > slab) slab = (min_capacity + 255) / 256 * 256;
        if (slab > TILE_SORT_BIG) slab = TILE_SORT_BIG;
        if (Tn * slab > 0x7FFFFFFFull || Tn * slab * 12 > SLAB_MAX_BYTES) {
            // slabs would not fit the 32-bit list positions / a sane scratch size: global sort instead
            return forward_impl(stream, geometry_buffer, geometry_user, binning_buffer, binning_user, image_buffer, image_user, P, D, M,
                                background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size, prefiltered, out_color,
                                out_coord, out_mcoord, out_depth, out_mdepth, out_alpha, out_normal, radii, require_coord, require_depth,
                                debug, true, 0, ex);
        }
        slab_size = (uint32_t)slab;
        const SlabLayout KL(Tn, slab);
        char* bbase = binning_buffer(binning_user, KL.total);
        if (!bbase) return fail(IGS_RAST_E_ALLOC, "binning buffer callback returned NULL");
        bbase = align_ptr(bbase);
        point_list = (uint32_t*)(bbase + KL.point_list);
        uint64_t* pairs = (uint64_t*)(bbase + KL.pairs);
        counters = (uint32_t*)(ibase + IL.counters);
        // The tile sort leaves the fill cursors and the count shards zeroed behind it, so a buffer that was clean before a frame is clean
        // after it: a caller that vouches for its buffer (igs_refine_step_args::scratch_clean) skips the zero-fill launch, unless
        // something on this thread was cut short between a binning kernel and its tile sort (g_counters_dirty).
        if (ex.scratch_clean && !g_counters_dirty) fp.zero_stats = stats;
        else HIP_TRY(zero_fill_async(s, tile_count, IL.zero_end - IL.tile_count), "zero tile counters");   // tile_count + stats + counters
        g_counters_dirty = true;
        prof_mark(s, ST_GAP);
        HIP_TRY(launch_preprocess_fwd(s, fp, rec, tiles, nullptr, nullptr, radii, counters, nullptr, 0, tile_count, pairs, slab_size),
                "preprocess_fwd launch");
        DBG_SYNC("preprocess_fwd");
        prof_mark(s, ST_PREPROCESS);
        // fused refine step: the tile sort also writes the blend kernel's dispatch order and leaves the fill cursors for that kernel to zero
        static const bool no_step_order = getenv("IGS_NO_STEP_ORDER") != nullptr;      // (A/B switch for measurements)
        use_step_order = !no_step_order && ex.fused_bwd && ex.skip_bwd_state && !colors_precomp && (require_coord != 0) == (require_depth != 0)
                         && step_order_usable((uint32_t)gx, (uint32_t)gy, slab_size);
        step_cursors = tile_count;
        HIP_TRY(launch_tile_sort(s, (uint32_t)Tn, tile_count, pairs, point_list, ranges, slab_size, stats, counters, (uint32_t)P,
                                 use_step_order ? (uint32_t*)(ibase + IL.tile_order) : nullptr, (uint32_t)gx, (uint32_t)gy), "tile_sort launch");
        g_counters_dirty = use_step_order;          // (then clean only once the blend kernel below has been launched)
        DBG_SYNC("tile_sort");
        prof_mark(s, ST_TILE_SORT);
        slab_stats = stats;
        g_last_fwd.overflow = stats + 1; g_last_fwd.prefilter = stats + 2;      // (the tile sort moved the flag there)
        slab_pending = true;
    } else {
        // ---------------- global radix binning (fallback for tiles denser than TILE_SORT_BIG) ----------------
        HIP_TRY(hipMemsetAsync(counters, 0, counter_bytes, s), "memset counters");

        uint32_t dnb = 0, dper = 0;
        sort_geometry((uint32_t)P, &dnb, &dper);
        HIP_TRY(hipMemsetAsync(ghist, 0, (size_t)256 * SORT_MAX_BLOCKS * 4, s), "memset depth-sort histogram 0");
        prof_mark(s, ST_GAP);
        HIP_TRY(launch_preprocess_fwd(s, fp, rec, tiles, keys_a, vals_a, radii, counters, ghist, dper, nullptr, nullptr, 0), "preprocess_fwd launch");
        DBG_SYNC("preprocess_fwd");
        prof_mark(s, ST_PREPROCESS);
        // instance count: read back while the depth sort runs
        HIP_TRY(hipMemcpyAsync(g_slot.pinned, counters, counter_bytes, hipMemcpyDeviceToHost, s), "memcpy count");
        HIP_TRY(hipEventRecord(g_slot.ev, s), "event record");
        prof_mark(s, ST_GAP);

        uint32_t *dk = nullptr, *order = nullptr;
        HIP_TRY(radix_sort_pairs(s, (uint32_t)P, keys_a, keys_b, vals_a, vals_b, ghist, 0, 32, &dk, &order), "depth sort launch");
        DBG_SYNC("depth sort");
        prof_mark(s, ST_DEPTH_SORT);
        const int nblk = (P + 255) / 256;
        HIP_TRY(launch_count_sorted(s, P, order, tiles, blocksum), "count_sorted launch");
        HIP_TRY(launch_scan_blocksums(s, nblk, blocksum), "scan_blocksums launch");
        DBG_SYNC("scan");
        prof_mark(s, ST_SCAN);

        HIP_TRY(hipEventSynchronize(g_slot.ev), "event sync");
        uint64_t R64 = 0;
        for (int sh = 0; sh < COUNTER_SHARDS; sh++) R64 += g_slot.pinned[COUNTER_SHARD_STRIDE * (1 + sh)];
        if (R64 > 0x7FFFFFFFull) return fail(IGS_RAST_E_INVALID, "instance count overflows int");
        R = (uint32_t)R64;
        if (g_slot.pinned[1]) return fail(IGS_RAST_E_PREFILTER, "Point is filtered although prefiltered is set. This shouldn't happen!");

        const BinLayout BL(R);
        char* bbase = binning_buffer(binning_user, BL.total);
        if (!bbase) return fail(IGS_RAST_E_ALLOC, "binning buffer callback returned NULL");
        bbase = align_ptr(bbase);
        point_list = (uint32_t*)(bbase + BL.point_list);
        uint32_t* bkeys_a = (uint32_t*)(bbase + BL.keys_a); uint32_t* bkeys_b = (uint32_t*)(bbase + BL.keys_b);
        uint32_t* bvals_b = (uint32_t*)(bbase + BL.vals_b);
        uint32_t* bhist = (uint32_t*)(bbase + BL.hist);

        HIP_TRY(hipMemsetAsync(ranges, 0, Tn * 8, s), "memset ranges");               // rasterizer_impl.cu:383
        prof_mark(s, ST_GAP);
        if (R > 0) {
            const int bits = ceil_log2((uint32_t)Tn);
            const int passes = (bits + 7) / 8;
            // arrange the ping-pong so that the sorted ids land in point_list
            uint32_t *ka, *kb, *va, *vb;
            if (passes % 2 == 0) { ka = bkeys_a; va = point_list; kb = bkeys_b; vb = bvals_b; }
            else                 { ka = bkeys_b; va = bvals_b;   kb = bkeys_a; vb = point_list; }
            uint32_t tnb = 0, tper = 0;
            sort_geometry(R, &tnb, &tper);
            HIP_TRY(hipMemsetAsync(bhist, 0, (size_t)256 * SORT_MAX_BLOCKS * 4, s), "memset tile-sort histogram 0");
            const uint32_t mask0 = bits >= 8 ? 255u : ((1u << bits) - 1u);
            HIP_TRY(launch_emit_instances(s, P, gx, gy, order, tiles, blocksum, rec, radii, ka, va, passes ? bhist : nullptr, tper, mask0),
                    "emit launch");
            DBG_SYNC("emit");
            prof_mark(s, ST_EMIT);
            uint32_t *sk = nullptr, *sv = nullptr;
            HIP_TRY(radix_sort_pairs(s, R, ka, kb, va, vb, bhist, 0, bits, &sk, &sv), "tile sort launch");
            DBG_SYNC("tile sort");
            prof_mark(s, ST_TILE_SORT);
            if (sv != point_list) return fail(IGS_RAST_E_INVALID, "internal: sort ping-pong mismatch");
            HIP_TRY(launch_tile_ranges(s, R, sk, ranges), "tile_ranges launch");
            DBG_SYNC("tile_ranges");
            prof_mark(s, ST_RANGES);
        }
    }

    BlendFwdArgs ba;
    ba.W = width; ba.H = height; ba.gx = gx; ba.gy = gy; ba.fx = fp.fx; ba.fy = fp.fy; ba.bg = background;
    ba.ranges = ranges; ba.point_list = point_list; ba.rec = rec; ba.colors_precomp = colors_precomp;
    ba.out_color = out_color; ba.out_coord = out_coord; ba.out_mcoord = out_mcoord; ba.out_depth = out_depth;
    ba.out_mdepth = out_mdepth; ba.out_alpha = out_alpha; ba.out_normal = out_normal;
    ba.n_contrib = (uint32_t*)(ibase + IL.n_contrib);
    ba.accum_coord = (float*)(ibase + IL.accum_coord); ba.accum_depth = (float*)(ibase + IL.accum_depth);
    ba.normal_length = (float*)(ibase + IL.normal_length);
    ba.stats_src = slab_stats; ba.flag_src = slab_stats ? slab_stats + 2 : counters + 1; ba.host_dst = slab_pending ? g_slot.pinned_dev : nullptr;
    ba.tile_order = (uint32_t*)(ibase + IL.tile_order);          // built on the side for the backward (both binning paths)
    ba.skip_bwd_state = ex.skip_bwd_state ? 1 : 0;
    if (slab_pending) { g_host_seq = g_host_seq + 1 ? g_host_seq + 1 : 1; g_slot.pinned[3] = 0; }
    ba.host_seq = g_host_seq;
    if (slab_pending && ex.no_latch) g_nowait_seq = g_host_seq;
    g_status_stream = s;
    const bool fuse_tiles = ex.fused_bwd && ex.skip_bwd_state && slab_pending && !colors_precomp && (require_coord != 0) == (require_depth != 0);
    if (ex.fused_ran) *ex.fused_ran = fuse_tiles;
    if (fuse_tiles) {
        BlendBwdArgs& bb = *ex.fused_bwd;
        bb.W = width; bb.H = height; bb.gx = gx; bb.gy = gy; bb.fx = fp.fx; bb.fy = fp.fy; bb.bg = background;
        bb.ranges = ranges; bb.point_list = point_list; bb.rec = rec; bb.colors_precomp = nullptr;
        bb.tile_order = nullptr;
        ba.tile_order = nullptr;                 // (no separate backward kernel that could use a load order)
        if (use_step_order) { ba.step_order = (const uint32_t*)(ibase + IL.tile_order); ba.reset_cursors = step_cursors; }
        HIP_TRY(launch_blend_step(s, ba, bb, require_coord != 0, require_depth != 0, ex.fused_instance), "blend_step launch");
        if (use_step_order) g_counters_dirty = false;
    } else {
        if (use_step_order) {                    // (cannot happen: the two conditions are the same terms -- but the cursors must not stay dirty)
            HIP_TRY(zero_fill_async(s, step_cursors, (size_t)Tn * 4), "zero tile counters");
            g_counters_dirty = false;
        }
        HIP_TRY(launch_blend_fwd(s, ba, require_coord != 0, require_depth != 0), "blend_fwd launch");
    }
    DBG_SYNC("blend_fwd");
    prof_mark(s, ST_BLEND_FWD);
    if (slab_pending && ex.defer_status) {
        if (!ex.no_latch) { g_pending.active = true; g_pending.slab = slab_size; }
        if (g_prof.on) g_prof.calls++;
        return 0x7FFFFFFF;                       // "unknown yet": an upper bound that igs_rast_backward accepts as R
    }
    if (slab_pending) {
        // only now does the host look at R: the whole pipeline above was enqueued without waiting for it
        if (int rc = wait_status(s)) return rc;
        const uint32_t R_dev = g_slot.pinned[0], overflow = g_slot.pinned[1];
        if (g_slot.pinned[2]) return fail(IGS_RAST_E_PREFILTER, "Point is filtered although prefiltered is set. This shouldn't happen!");
        if (R_dev > 0x7FFFFFFFu) return fail(IGS_RAST_E_INVALID, "instance count overflows int");
        if (overflow) {
            // a tile holds more instances than its slab (overflow = the largest such tile): redo with what is now known
            const bool radix = overflow > (uint32_t)TILE_SORT_BIG;
            const uint64_t want = ((uint64_t)overflow + overflow / 4 + 255) / 256 * 256;
            g_hint.slab = (uint32_t)(want > TILE_SORT_BIG ? TILE_SORT_BIG : want);
            return forward_impl(stream, geometry_buffer, geometry_user, binning_buffer, binning_user, image_buffer, image_user, P, D, M,
                                background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size, prefiltered, out_color,
                                out_coord, out_mcoord, out_depth, out_mdepth, out_alpha, out_normal, radii, require_coord, require_depth,
                                debug, radix, overflow, ex);
        }
        R = R_dev;
    }
    if (g_prof.on) { g_prof.r_sum += (double)R; g_prof.calls++; }
    return (int)R;
}

extern "C" int igs_rast_forward(
    void* stream,
    igs_rast_alloc_fn geometry_buffer, void* geometry_user, igs_rast_alloc_fn binning_buffer, void* binning_user,
    igs_rast_alloc_fn image_buffer, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
    float* out_color, float* out_coord, float* out_mcoord, float* out_depth, float* out_mdepth, float* out_alpha,
    float* out_normal, int* radii, int require_coord, int require_depth, int debug)
{
    const bool hint_clean = g_hint_clean; g_hint_clean = false;      // one-shot promise: consumed by THIS call, even a refused one
    if (g_pending.active) return fail(IGS_RAST_E_INVALID, "igs_rast_forward: an asynchronous forward is pending on this thread; call igs_rast_forward_finish() first");
    prof_new_frame();
    const char* e = getenv("IGS_BINNING");                    // "radix" forces the global-sort path (tests)
    const bool radix = e && strcmp(e, "radix") == 0;
    FwdExtra ex0; ex0.scratch_clean = hint_clean;
    return forward_impl(stream, geometry_buffer, geometry_user, binning_buffer, binning_user, image_buffer, image_user, P, D, M,
                        background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                        cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size, prefiltered, out_color,
                        out_coord, out_mcoord, out_depth, out_mdepth, out_alpha, out_normal, radii, require_coord, require_depth,
                        debug, radix, 0, ex0);
}

// Asynchronous variant for callers that keep enqueueing work (the native refine step): identical to igs_rast_forward but
// returns right after the last launch WITHOUT waiting for the instance count; the return value is INT_MAX ("not known
// yet", which igs_rast_backward accepts as R).  igs_rast_forward_finish() then waits for the small
// read-back (which completed right after the tile scan, long before the blend) and returns the true num_rendered, or
// IGS_RAST_E_RETRY if a tile overflowed its instance slab: everything enqueued since must then be discarded and
// the frame redone with igs_rast_forward (the hints are updated, so it will fit).
extern "C" int igs_rast_forward_async(
    void* stream,
    igs_rast_alloc_fn geometry_buffer, void* geometry_user, igs_rast_alloc_fn binning_buffer, void* binning_user,
    igs_rast_alloc_fn image_buffer, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
    float* out_color, float* out_coord, float* out_mcoord, float* out_depth, float* out_mdepth, float* out_alpha,
    float* out_normal, int* radii, int require_coord, int require_depth, int debug)
{
    // the status slot is single: a second frame must not be started before the first one's count has been collected
    const bool hint_clean = g_hint_clean; g_hint_clean = false;      // (consumed by this call, even a refused one)
    if (g_pending.active) return fail(IGS_RAST_E_INVALID, "igs_rast_forward_async: the previous asynchronous forward has not been finished (igs_rast_forward_finish)");
    prof_new_frame();
    FwdExtra ex; ex.defer_status = true; ex.scratch_clean = hint_clean;
    const int rc = forward_impl(stream, geometry_buffer, geometry_user, binning_buffer, binning_user, image_buffer, image_user, P, D, M,
                                background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size, prefiltered, out_color,
                                out_coord, out_mcoord, out_depth, out_mdepth, out_alpha, out_normal, radii, require_coord, require_depth,
                                debug, false, 0, ex);
    return rc;
}
extern "C" int igs_rast_forward_finish(void)
{
    if (!g_pending.active) return fail(IGS_RAST_E_INVALID, "igs_rast_forward_finish: no asynchronous forward pending");
    g_pending.active = false;
    if (int rc = wait_status(g_status_stream)) return rc;
    const uint32_t R_dev = g_slot.pinned[0], overflow = g_slot.pinned[1];
    if (g_slot.pinned[2]) return fail(IGS_RAST_E_PREFILTER, "Point is filtered although prefiltered is set. This shouldn't happen!");
    if (R_dev > 0x7FFFFFFFu) return fail(IGS_RAST_E_INVALID, "instance count overflows int");
    if (overflow) {
        const uint64_t want = ((uint64_t)overflow + overflow / 4 + 255) / 256 * 256;
        g_hint.slab = (uint32_t)(want > TILE_SORT_BIG ? TILE_SORT_BIG : want);       // igs_rast_forward falls back further if needed
        return fail(IGS_RAST_E_RETRY, "a tile overflowed its instance slab: redo the frame");
    }
    if (g_prof.on) g_prof.r_sum += (double)R_dev;
    return (int)R_dev;
}

// Forward for stream capture (hipGraph): identical launches, NO host-side wait, no pending latch -- nothing in it is illegal
// while the stream is capturing (the pinned status slot must exist already: run one ordinary forward on this thread and device
// first).  Every replay of the captured launches posts its {R, overflow, prefilter flag} into the slot;
// igs_rast_last_status() reads it back once the caller has synchronised the stream.
extern "C" int igs_rast_forward_nowait(
    void* stream,
    igs_rast_alloc_fn geometry_buffer, void* geometry_user, igs_rast_alloc_fn binning_buffer, void* binning_user,
    igs_rast_alloc_fn image_buffer, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
    float* out_color, float* out_coord, float* out_mcoord, float* out_depth, float* out_mdepth, float* out_alpha,
    float* out_normal, int* radii, int require_coord, int require_depth, int debug)
{
    const bool hint_clean = g_hint_clean; g_hint_clean = false;      // (consumed by this call, even a refused one)
    if (g_pending.active) return fail(IGS_RAST_E_INVALID, "igs_rast_forward_nowait: an asynchronous forward is pending on this thread; call igs_rast_forward_finish() first");
    if (debug) return fail(IGS_RAST_E_INVALID, "igs_rast_forward_nowait: debug (a synchronisation after every launch) cannot be captured");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= IGS_MAX_DEVICES || !g_slots.slot[dev].pinned)
        return fail(IGS_RAST_E_INVALID, "igs_rast_forward_nowait: no status slot on this thread and device yet (run one igs_rast_forward first: pinned memory cannot be allocated during capture)");
    prof_new_frame();
    FwdExtra ex; ex.defer_status = true; ex.no_latch = true; ex.scratch_clean = hint_clean;
    return forward_impl(stream, geometry_buffer, geometry_user, binning_buffer, binning_user, image_buffer, image_user, P, D, M,
                        background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                        cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size, prefiltered, out_color,
                        out_coord, out_mcoord, out_depth, out_mdepth, out_alpha, out_normal, radii, require_coord, require_depth,
                        0, false, 0, ex);
}
// What the last slab-binned forward of this thread and device posted.  *overflow != 0: a tile needed that many instance slots and
// the per-tile slabs were smaller -- the frame (and everything computed from it) is invalid; the slab hint has been raised, so a
// new capture / an ordinary igs_rast_forward will fit.  Only meaningful once the stream has been synchronised.
static int last_status(int* num_rendered, unsigned* overflow, unsigned* prefilter_flag, bool check_seq)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= IGS_MAX_DEVICES || !g_slots.slot[dev].pinned)
        return fail(IGS_RAST_E_INVALID, "igs_rast_last_status: no forward has run on this thread and device");
    const uint32_t* p = g_slots.slot[dev].pinned;
    // the sequence word is stored last by the kernel (release): if it is not the number baked into the last igs_rast_forward_nowait of this
    // thread (the last capture), no replay of that capture has posted yet and the other three words still belong to an EARLIER (eager) frame
    const uint32_t seq = __atomic_load_n(&p[3], __ATOMIC_ACQUIRE);
    if (check_seq && seq != (g_nowait_seq ? g_nowait_seq : g_host_seq))
        return fail(IGS_RAST_E_RETRY, "igs_rast_last_status: the captured forward has not posted its status yet (replay the graph and synchronise the stream first)");
    const uint32_t R = __atomic_load_n(&p[0], __ATOMIC_ACQUIRE), ov = __atomic_load_n(&p[1], __ATOMIC_ACQUIRE);
    if (num_rendered) *num_rendered = R > 0x7FFFFFFFu ? 0x7FFFFFFF : (int)R;
    if (overflow) *overflow = ov;
    if (prefilter_flag) *prefilter_flag = p[2];
    if (ov) {
        const uint64_t want = ((uint64_t)ov + ov / 4 + 255) / 256 * 256;
        if (want > g_hint.slab) g_hint.slab = (uint32_t)(want > TILE_SORT_BIG ? TILE_SORT_BIG : want);
    }
    return 0;
}
extern "C" int igs_rast_last_status(int* num_rendered, unsigned* overflow, unsigned* prefilter_flag)
{
    return last_status(num_rendered, overflow, prefilter_flag, true);
}
extern "C" int igs_rast_last_posted_status(int* num_rendered, unsigned* overflow, unsigned* prefilter_flag)
{
    return last_status(num_rendered, overflow, prefilter_flag, false);
}

extern "C" void igs_rast_set_slab_hint(unsigned slots_per_tile) { g_hint.slab = slots_per_tile > TILE_SORT_BIG ? TILE_SORT_BIG : slots_per_tile; }
extern "C" unsigned igs_rast_get_slab_hint(void) { return g_hint.slab; }

// which blend_bwd_kernel<COORD, DEPTH, NORMAL, ABS> the last backward of this PROCESS launched (not per thread: PyTorch runs
// autograd backward functions on its own worker thread): bit 0 coord, 1 depth, 2 normal, 3 abs-gradient moment; -1 = none
// (R == 0 / no backward yet).  Tests use it to prove that absent upstream gradients select the cheaper instance.
static int g_last_bwd_instance = -1;
extern "C" int igs_rast_last_backward_instance(void) { return __atomic_load_n(&g_last_bwd_instance, __ATOMIC_RELAXED); }

// NaN report of the per-Gaussian backward kernel (replaces the reference's seven `assert not torch.isnan(g).any()` host syncs,
// DGR/diff_gaussian_rasterization_rade/__init__.py:156-162): a thread that writes a NaN stores the report's sequence number into a word
// of pinned host memory; an event recorded behind the kernel tells the host when the word is final.  One TICKET per report, a ring of
// NAN_RING per host thread and device (two renders in one backward pass, ...): memory and events are never freed -- the thread that
// runs autograd backward functions may outlive the HIP runtime.
#define NAN_RING 256
struct NanTicket { volatile uint32_t* word = nullptr; hipEvent_t ev = nullptr; uint32_t seq = 0; };
struct NanSlot { uint32_t* pinned = nullptr; uint32_t* pinned_dev = nullptr; NanTicket* tickets = nullptr; uint32_t seq = 0; bool requested = false; bool pending = false; };
static thread_local NanSlot g_nan[IGS_MAX_DEVICES];
static thread_local int g_nan_dev = -1;          // device of the last backward that was asked for a report
static thread_local float g_next_clamp = 0.f;     // one-shot: clamp of the NEXT igs_rast_backward of this thread (clamp package)
extern "C" void igs_rast_next_backward_options(int nan_report, float clamp_grads)
{
    g_next_clamp = clamp_grads > 0.f ? clamp_grads : 0.f;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= IGS_MAX_DEVICES) return;
    g_nan[dev].requested = nan_report != 0;
}
static int nan_wait_ticket(const NanTicket* t, uint32_t seq)
{
    if (!t || !t->ev) return fail(IGS_RAST_E_INVALID, "NaN report: bad ticket");
    double t0 = 0.0;
    for (long spins = 0;; spins++) {
        if (t->seq != seq) return fail(IGS_RAST_E_INVALID, "the NaN report was overwritten by a later one before it was read");
        const hipError_t q = hipEventQuery(t->ev);
        if (q == hipSuccess) return (__atomic_load_n(t->word, __ATOMIC_ACQUIRE) == seq) ? 1 : 0;
        if (q != hipErrorNotReady) return fail(IGS_RAST_E_HIP, "error while waiting for the NaN report", q);
        if ((spins & 0xFF) == 0xFF) {
            const double now = now_s();
            if (t0 == 0.0) t0 = now;
            else if (now - t0 > wait_limit_s()) return fail(IGS_RAST_E_HIP, "timed out waiting for the NaN report (IGS_RAST_WAIT_TIMEOUT_S)");
        }
    }
}
extern "C" int igs_rast_nan_report_wait(void)
{
    if (g_nan_dev < 0 || !g_nan[g_nan_dev].pending) return fail(IGS_RAST_E_INVALID, "igs_rast_nan_report_wait: no backward with a NaN report pending on this thread");
    NanSlot& n = g_nan[g_nan_dev];
    n.pending = false;
    return nan_wait_ticket(&n.tickets[n.seq % NAN_RING], n.seq);
}
extern "C" int igs_rast_nan_report_handle(const void** ticket, unsigned* seq)
{
    if (g_nan_dev < 0 || !g_nan[g_nan_dev].pending || !ticket || !seq) return fail(IGS_RAST_E_INVALID, "igs_rast_nan_report_handle: no backward with a NaN report pending on this thread");
    NanSlot& n = g_nan[g_nan_dev];
    n.pending = false;
    *ticket = &n.tickets[n.seq % NAN_RING]; *seq = n.seq;
    return 0;
}
extern "C" int igs_rast_nan_report_wait_at(const void* ticket, unsigned seq)
{
    if (!ticket) return fail(IGS_RAST_E_INVALID, "igs_rast_nan_report_wait_at: NULL handle");
    return nan_wait_ticket((const NanTicket*)ticket, seq);
}

// l1_gt != NULL: L1 loss fused into the blend backward (dL_dpix ignored); fuse != NULL: activation backward + Adam fused into
// the per-Gaussian backward (no gradient outputs except the optional dL_dmean2D).
static int backward_impl(
    void* stream, int P, int D, int M, int R, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, float kernel_size, const int* radii, const float* normalmap,
    const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
    const float* dL_dpix, const float* dL_dpix_coord, const float* dL_dpix_mcoord, const float* dL_dpix_depth,
    const float* dL_dpix_mdepth, const float* dL_dalphas, const float* dL_dpixel_normals,
    void* workspace,
    float* dL_dmean2D, float* dL_dcolor, float* dL_dopacity, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
    float* dL_dscale, float* dL_drot, int require_coord, int require_depth, int debug,
    const float* l1_gt, const float* l1_color, float l1_scale, const RefineFuse* fuse)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(IGS_RAST_E_INVALID, "igs_rast_backward: bad sizes");
    if (P == 0) return 0;                                       // rasterize_points.cu:195
    if (!geom_buffer || !image_buffer || (!binning_buffer && R > 0) || !workspace)
        return fail(IGS_RAST_E_INVALID, "igs_rast_backward: NULL scratch buffer");
    if (!means3D || !alphas || !viewmatrix || !projmatrix || !campos || !background || !radii || !normalmap)
        return fail(IGS_RAST_E_INVALID, "igs_rast_backward: NULL required input");
    // any of the seven upstream gradients may be NULL = "all zeros" (an output that did not take part in the loss)
    if (!fuse && (!dL_dmean2D || !dL_dcolor || !dL_dopacity || !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot || (M > 0 && !dL_dsh)))
        return fail(IGS_RAST_E_INVALID, "igs_rast_backward: NULL output");

    if ((uint64_t)P * GACC_F * 4 >= (1ull << 32))      // (the blend backward addresses its accumulator rows with 32-bit byte offsets)
        return fail(IGS_RAST_E_INVALID, "igs_rast_backward: more than 33 million Gaussians are not supported");
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    const size_t Tn = (size_t)gx * gy, HW = (size_t)width * height;
    const GeomLayout GL(P);
    const ImgLayout IL(HW, Tn);
    const BinLayout BL(R);
    const char* gbase = align_ptr(geom_buffer);
    const char* ibase = align_ptr(image_buffer);
    const char* bbase = binning_buffer ? align_ptr(binning_buffer) : nullptr;
    float* gacc = (float*)align_ptr((const char*)workspace);
    const float fy = height / (2.0f * tan_fovy), fx = width / (2.0f * tan_fovx);

    // NaN report asked for this backward (one-shot, consumed here even if the call fails later)
    NanSlot* nan = nullptr;
    const float clamp_next = g_next_clamp; g_next_clamp = 0.f;
    if (!fuse) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < IGS_MAX_DEVICES && g_nan[dev].requested) {
            g_nan[dev].requested = false;
            NanSlot& n = g_nan[dev];
            if (!n.pinned) {
                HIP_TRY(hipHostMalloc((void**)&n.pinned, NAN_RING * 4, hipHostMallocDefault), "hipHostMalloc");
                HIP_TRY(hipHostGetDevicePointer((void**)&n.pinned_dev, n.pinned, 0), "hipHostGetDevicePointer");
                memset(n.pinned, 0, NAN_RING * 4);
                n.tickets = new NanTicket[NAN_RING];
            }
            nan = &n; g_nan_dev = dev;
        }
    }
    prof_mark(s, ST_GAP);
    float* loss_shards = (float*)((char*)gacc + ws_gacc_bytes(P));
    // which blend instance will run (launch_blend_bwd decides the same way): the colour-only one packs its moments into 64-byte rows
    const bool will_compact = !(require_coord && (dL_dpix_coord || dL_dpix_mcoord)) && !(require_depth && (dL_dpix_depth || dL_dpix_mdepth))
                              && !((require_coord || require_depth) && dL_dpixel_normals);
    if (!(fuse && fuse->prezeroed)) {              // (igs_refine_step: the forward's preprocess kernel zero-filled the accumulators)
        HIP_TRY(zero_fill_async(s, gacc, (size_t)P * (will_compact ? GACC_COMPACT_F : GACC_F) * 4), "zero gacc");
        if (l1_gt) HIP_TRY(zero_fill_async(s, loss_shards, WS_LOSS_BYTES), "zero loss shards");
    }
    prof_mark(s, ST_MEMSET);
    BlendBwdArgs ba;
    ba.W = width; ba.H = height; ba.gx = gx; ba.gy = gy; ba.fx = fx; ba.fy = fy; ba.bg = background;
    ba.ranges = (const uint32_t*)(ibase + IL.ranges);
    ba.point_list = bbase ? (const uint32_t*)(bbase + BL.point_list) : nullptr;
    ba.rec = (const float*)(gbase + GL.rec); ba.colors_precomp = colors_precomp;
    ba.alphas = alphas; ba.normalmap = normalmap;
    ba.accum_coord = (const float*)(ibase + IL.accum_coord); ba.accum_depth = (const float*)(ibase + IL.accum_depth);
    ba.normal_length = (const float*)(ibase + IL.normal_length); ba.n_contrib = (const uint32_t*)(ibase + IL.n_contrib);
    ba.dL_dpix = dL_dpix; ba.dL_dcoord = dL_dpix_coord; ba.dL_dmcoord = dL_dpix_mcoord; ba.dL_ddepth = dL_dpix_depth;
    ba.dL_dmdepth = dL_dpix_mdepth; ba.dL_dalpha = dL_dalphas; ba.dL_dnormal = dL_dpixel_normals;
    ba.gacc = gacc;
    ba.l1_gt = l1_gt; ba.l1_color = l1_color; ba.l1_scale = l1_scale; ba.l1_loss = loss_shards;
    ba.want_absgrad = (fuse && !dL_dmean2D) ? 0 : 1;
    ba.tile_order = (const uint32_t*)(ibase + IL.tile_order);    // the forward's blend kernel ordered the tiles heaviest-first
    bool gacc_compact = will_compact;
    int inst_bits = -1;
    if (fuse && fuse->blend_done) {
        // the blend backward ran inside the forward's tile kernel (blend_step.hip): colour-only moments in compact rows
        if (!will_compact) return fail(IGS_RAST_E_INVALID, "internal: fused tile kernel with a non-compact accumulator layout");
        prof_mark(s, ST_BLEND_BWD);
    } else if (R > 0) {
        HIP_TRY(launch_blend_bwd(s, ba, require_coord != 0, require_depth != 0, &gacc_compact, &inst_bits), "blend_bwd launch");
        if (gacc_compact != will_compact) return fail(IGS_RAST_E_INVALID, "internal: accumulator layout mismatch");
        __atomic_store_n(&g_last_bwd_instance, inst_bits, __ATOMIC_RELAXED);
        DBG_SYNC("blend_bwd");
        prof_mark(s, ST_BLEND_BWD);
    } else __atomic_store_n(&g_last_bwd_instance, -1, __ATOMIC_RELAXED);
    GeomBwdArgs ga;
    ga.P = P; ga.D = D; ga.M = shs ? M : 0; ga.W = width; ga.H = height;
    ga.means3D = means3D; ga.shs = shs; ga.scales = scales; ga.rotations = rotations; ga.cov3D_precomp = cov3D_precomp;
    ga.radii = radii; ga.scale_modifier = scale_modifier; ga.tan_fovx = tan_fovx; ga.tan_fovy = tan_fovy;
    ga.fx = fx; ga.fy = fy; ga.kernel_size = kernel_size;
    ga.view = viewmatrix; ga.proj = projmatrix; ga.campos = campos;
    ga.rec = ba.rec; ga.gacc = gacc; ga.gacc_compact = gacc_compact ? 1 : 0;
    if (fuse && fuse->plane_tag) { ga.plane_cache = (const float*)(gbase + GL.planes); ga.plane_tag = fuse->plane_tag; }
    ga.dL_dmean2D = dL_dmean2D; ga.dL_dcolor = dL_dcolor; ga.dL_dopacity = dL_dopacity; ga.dL_dmean3D = dL_dmean3D;
    ga.dL_dcov3D = dL_dcov3D; ga.dL_dsh = dL_dsh; ga.dL_dscale = dL_dscale; ga.dL_drot = dL_drot;
    if (fuse) {
        RefineFuse f = *fuse;
        if (!f.loss_shards) f.loss_shards = loss_shards;
        if (f.color_event && f.color_out && f.grad_out) {
            // N > 1: this view's colour gradients leave one kernel early, the all-gather runs underneath the per-Gaussian kernel
            HIP_TRY(launch_extract_view_colors(s, P, radii, ba.rec, gacc, gacc_compact ? 1 : 0, shs != nullptr && M > 0, f.color_out), "extract_view_colors launch");
            HIP_TRY(hipEventRecord((hipEvent_t)f.color_event, s), "record colour event");
            f.color_out = nullptr; f.colors_extracted = 1;
        }
        HIP_TRY(launch_geom_bwd_adam(s, ga, f), "geom_bwd_adam launch");
    } else {
        ga.clamp = clamp_next;
        NanTicket* ticket = nullptr;
        if (nan) {
            nan->seq = nan->seq + 1 ? nan->seq + 1 : 1;
            ticket = &nan->tickets[nan->seq % NAN_RING];
            if (!ticket->ev) HIP_TRY(hipEventCreateWithFlags(&ticket->ev, hipEventDisableTiming), "hipEventCreate");
            ticket->word = nan->pinned + (nan->seq % NAN_RING); ticket->seq = nan->seq;
            ga.nan_host = nan->pinned_dev + (nan->seq % NAN_RING); ga.nan_seq = nan->seq;
        }
        HIP_TRY(launch_geom_bwd(s, ga), "geom_bwd launch");
        if (ticket) { HIP_TRY(hipEventRecord(ticket->ev, s), "record NaN-report event"); nan->pending = true; }
    }
    DBG_SYNC("geom_bwd");
    prof_mark(s, ST_GEOM_BWD);
    return 0;
}

extern "C" int igs_rast_backward(
    void* stream, int P, int D, int M, int R, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* alphas,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tan_fovx, float tan_fovy, float kernel_size, const int* radii, const float* normalmap,
    const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
    const float* dL_dpix, const float* dL_dpix_coord, const float* dL_dpix_mcoord, const float* dL_dpix_depth,
    const float* dL_dpix_mdepth, const float* dL_dalphas, const float* dL_dpixel_normals,
    void* workspace,
    float* dL_dmean2D, float* dL_dcolor, float* dL_dopacity, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
    float* dL_dscale, float* dL_drot, int require_coord, int require_depth, int debug)
{
    return backward_impl(stream, P, D, M, R, background, width, height, means3D, shs, colors_precomp, alphas, scales, scale_modifier,
                         rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, kernel_size, radii, normalmap,
                         geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dpix_coord, dL_dpix_mcoord, dL_dpix_depth,
                         dL_dpix_mdepth, dL_dalphas, dL_dpixel_normals, workspace, dL_dmean2D, dL_dcolor, dL_dopacity, dL_dmean3D,
                         dL_dcov3D, dL_dsh, dL_dscale, dL_drot, require_coord, require_depth, debug, nullptr, nullptr, 0.f, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// One refine iteration on one view, single GPU (infer_batch.py:279-324 with the L1 loss): activations -> render -> L1 ->
// backward -> Adam, 6 launches, no gradient array in HBM, no host wait before the last launch is enqueued.
// ---------------------------------------------------------------------------------------------------------------------
struct ScratchCapture { igs_rast_alloc_fn fn; void* user; char* last; };
static char* capture_alloc(void* user, size_t n) { ScratchCapture* c = (ScratchCapture*)user; c->last = c->fn(c->user, n); return c->last; }

extern "C" size_t igs_refine_loss_scratch_bytes(int width, int height)
{
    const size_t HW = (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0);
    // { SSIM scratch | dL/dcolor 3 HW | depth-normal: dL/ddepth HW, dL/dmdepth HW, dL/dnormal 3 HW | 64 shards }
    return ((igs_ssim_l1_scratch_bytes(width, height) + 255) & ~(size_t)255) + 3 * HW * 4 + 5 * HW * 4 + 4096 + 512;
}

extern "C" size_t igs_refine_step_args_size(void) { return sizeof(igs_refine_step_args); }

extern "C" int igs_refine_step(const igs_refine_step_args* a)
{
    if (!a) return fail(IGS_RAST_E_INVALID, "igs_refine_step: NULL args");
    if (g_pending.active) return fail(IGS_RAST_E_INVALID, "igs_refine_step: an asynchronous forward is pending on this thread; call igs_rast_forward_finish() first");
    if (a->P <= 0 || a->M <= 0 || a->width <= 0 || a->height <= 0 || (a->step < 1 && !a->grad_out))
        return fail(IGS_RAST_E_INVALID, "igs_refine_step: bad sizes");
    if (!a->param || (!a->grad_out && (!a->exp_avg || !a->exp_avg_sq)) || !a->gt || !a->out_images || !a->radii || !a->workspace || !a->background)
        return fail(IGS_RAST_E_INVALID, "igs_refine_step: NULL pointer");
    const size_t HW = (size_t)a->width * a->height;
    float* img = a->out_images;
    float *color = img, *coord = img + 3 * HW, *mcoord = img + 6 * HW, *depth = img + 9 * HW, *mdepth = img + 10 * HW,
          *alpha = img + 11 * HW, *normal = img + 12 * HW;
    const float* xyz = a->param + a->off_xyz; const float* shs = a->param + a->off_sh;
    const float* opac = a->param + a->off_opacity; const float* scal = a->param + a->off_scale; const float* rotn = a->param + a->off_rot;
    ScratchCapture cg{ a->geometry_buffer, a->geometry_user, nullptr }, cb{ a->binning_buffer, a->binning_user, nullptr },
                   ci{ a->image_buffer, a->image_user, nullptr };
    const int stepno = a->step < 1 ? 1 : a->step;
    const double bc1 = 1.0 - pow((double)a->beta1, (double)stepno), bc2 = 1.0 - pow((double)a->beta2, (double)stepno);
    RefineFuse f;
    f.param = a->param; f.exp_avg = a->exp_avg; f.exp_avg_sq = a->exp_avg_sq; f.grad_out = a->grad_out;
    f.off_xyz = a->off_xyz; f.off_rot = a->off_rot; f.off_sh = a->off_sh; f.off_opacity = a->off_opacity; f.off_scale = a->off_scale;
    f.lr_xyz = (float)(a->lr_xyz / bc1); f.lr_rot = (float)(a->lr_rot / bc1); f.lr_sh = (float)(a->lr_sh / bc1);
    f.lr_opacity = (float)(a->lr_opacity / bc1); f.lr_scale = (float)(a->lr_scale / bc1);
    f.clamp = a->clamp_grads;
    f.color_out = a->color_grad_out;
    f.color_event = a->color_ready_event;
    f.b1 = a->beta1; f.b2 = a->beta2; f.eps = a->eps; f.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const float inv_n = 1.0f / (float)(3 * HW);
    const bool dssim = a->lambda_dssim > 0.f;
    const bool dn = a->lambda_depth_normal > 0.f;
    if ((dssim || dn) && !a->loss_scratch) return fail(IGS_RAST_E_INVALID, "igs_refine_step: lambda_dssim / lambda_depth_normal > 0 need loss_scratch");
    if (dn && !a->require_depth) return fail(IGS_RAST_E_INVALID, "igs_refine_step: the depth-normal regulariser needs require_depth");
    // scratch of the SSIM mode: { loss kernels' own scratch (maps + 2 x 64 shards) | dL/dcolor [3][H][W] }
    float* ssim_shards = nullptr; float* grad_img = nullptr;
    float *dn_gd = nullptr, *dn_gm = nullptr, *dn_gn = nullptr, *dn_shards = nullptr;
    if (dn) {
        const size_t own = (igs_ssim_l1_scratch_bytes(a->width, a->height) + 255) & ~(size_t)255;
        float* base = (float*)((char*)a->loss_scratch + own) + 3 * HW;
        dn_gd = base; dn_gm = base + HW; dn_gn = base + 2 * HW; dn_shards = (float*)(((uintptr_t)(base + 5 * HW) + 255) & ~(uintptr_t)255);
    }
    if (dssim) {
        const size_t own = (igs_ssim_l1_scratch_bytes(a->width, a->height) + 255) & ~(size_t)255;
        ssim_shards = (float*)((char*)a->loss_scratch + (((size_t)9 * HW * 4 + 255) & ~(size_t)255));
        grad_img = (float*)((char*)a->loss_scratch + own);
    }
    f.loss_out = a->loss_out; f.loss_shards2 = nullptr; f.loss_bias = 0.f; f.loss_scale2 = 0.f;
    f.loss_shards3 = dn_shards; f.loss_scale3 = 1.0f;       // (the kernel's shards already carry weight * lambda / HW)
    if (dssim) {          // loss = lambda w (1 - mean ssim) + (1 - lambda) w mean|d|
        f.loss_shards = ssim_shards; f.loss_scale = -a->lambda_dssim * a->loss_weight * inv_n;
        f.loss_shards2 = ssim_shards + 1024; f.loss_scale2 = (1.f - a->lambda_dssim) * a->loss_weight * inv_n;
        f.loss_bias = a->lambda_dssim * a->loss_weight;
    } else {
        f.loss_shards = nullptr; f.loss_scale = a->loss_weight * inv_n;
    }
    const float l1_scale = a->loss_weight * inv_n;

    prof_new_frame();
    for (int attempt = 0; attempt < 2; attempt++) {
        g_pending.active = false;
        f.prezeroed = 1;
        FwdExtra ex;
        ex.zero_gacc = (float*)align_ptr((const char*)a->workspace);
        ex.zero_loss = dssim ? ssim_shards : (float*)((char*)ex.zero_gacc + ws_gacc_bytes(a->P));
        ex.zero_loss2 = dssim ? ssim_shards + 1024 : nullptr;
        ex.defer_status = attempt == 0;            // second attempt: synchronous forward, which sorts out its scratch sizes itself
        ex.raw_activations = true;
        ex.skip_bwd_state = !dn;                   // (colour-only backward instance: see BlendFwdArgs)
        static const bool no_plane_cache = getenv("IGS_NO_PLANE_CACHE") != nullptr;      // (A/B switch for measurements)
        if (dn && !no_plane_cache) {
            // the regulariser sends depth / normal gradients back: the forward keeps Sigma^-1 of every visible Gaussian, the
            // per-Gaussian backward takes it from there instead of running the eigen-solver again (geom_math.h: PlaneCache)
            static std::atomic<uint32_t> nonce{0};
            uint32_t t = ++nonce;
            if (t == 0) t = ++nonce;
            ex.plane_tag = t; f.plane_tag = t;
        } else f.plane_tag = 0;
        ex.scratch_clean = a->scratch_clean != 0;
        // L1 loss, colour-only backward: forward and backward blend of a tile in one kernel (blend_step.hip)
        BlendBwdArgs fused_bwd;
        bool fused_ran = false;
        int fused_inst = -1;
        static const bool no_fuse = getenv("IGS_NO_TILE_FUSION") != nullptr;      // (A/B switch for measurements)
        if (!dssim && !dn && !no_fuse) {
            fused_bwd.alphas = nullptr; fused_bwd.normalmap = nullptr; fused_bwd.accum_coord = nullptr; fused_bwd.accum_depth = nullptr;
            fused_bwd.normal_length = nullptr; fused_bwd.n_contrib = nullptr;
            fused_bwd.dL_dpix = nullptr; fused_bwd.dL_dcoord = nullptr; fused_bwd.dL_dmcoord = nullptr; fused_bwd.dL_ddepth = nullptr;
            fused_bwd.dL_dmdepth = nullptr; fused_bwd.dL_dalpha = nullptr; fused_bwd.dL_dnormal = nullptr;
            fused_bwd.gacc = ex.zero_gacc;
            fused_bwd.l1_gt = a->gt; fused_bwd.l1_color = color; fused_bwd.l1_scale = l1_scale;
            fused_bwd.l1_loss = (float*)((char*)ex.zero_gacc + ws_gacc_bytes(a->P));
            fused_bwd.want_absgrad = a->dL_dmean2D ? 1 : 0;
            ex.fused_bwd = &fused_bwd; ex.fused_ran = &fused_ran; ex.fused_instance = &fused_inst;
        }
        const int R = forward_impl(a->stream, capture_alloc, &cg, capture_alloc, &cb, capture_alloc, &ci, a->P, a->D, a->M, a->background,
                                   a->width, a->height, xyz, shs, nullptr, opac, scal, 1.0f, rotn, nullptr, a->viewmatrix, a->projmatrix,
                                   a->cam_pos, a->tan_fovx, a->tan_fovy, 0.0f, 0, color, coord, mcoord, depth, mdepth, alpha, normal,
                                   a->radii, a->require_coord, a->require_depth, 0, false, 0, ex);
        if (R < 0) return R;
        f.guard_overflow = g_last_fwd.overflow; f.guard_prefilter = g_last_fwd.prefilter;
        f.blend_done = fused_ran ? 1 : 0;
        if (fused_ran) __atomic_store_n(&g_last_bwd_instance, fused_inst, __ATOMIC_RELAXED);
        DepthNormalJob dnj;
        if (dn) {
            // depth-normal regulariser on the maps just rendered: its three gradient maps switch the blend backward to the
            // <depth, normal> instance
            dnj.fx = a->width / (2.0f * a->tan_fovx); dnj.fy = a->height / (2.0f * a->tan_fovy);
            dnj.depth = depth; dnj.mdepth = mdepth; dnj.normal = normal; dnj.weight = a->loss_weight * a->lambda_depth_normal;
            dnj.depth_ratio = a->depth_ratio > 0.f ? a->depth_ratio : 0.6f;
            dnj.g_depth = dn_gd; dnj.g_mdepth = dn_gm; dnj.g_normal = dn_gn; dnj.loss_shards = dn_shards;
        }
        static const bool no_mix = getenv("IGS_NO_LOSS_MIX") != nullptr;      // (A/B switch for measurements)
        const bool mix = dssim && dn && !no_mix;       // both image-space losses: the regulariser rides in the SSIM gradient's launch
        if (dssim) {
            prof_mark((hipStream_t)a->stream, ST_GAP);
            if (launch_ssim_l1((hipStream_t)a->stream, a->width, a->height, color, a->gt, a->lambda_dssim, a->loss_weight, a->loss_scratch,
                               grad_img, false, a->gt_stats, a->gt_stats_valid != 0, mix ? &dnj : nullptr) != hipSuccess)
                return fail(IGS_RAST_E_HIP, "ssim loss launch");
            prof_mark((hipStream_t)a->stream, ST_MEMSET);          // (the stage slot the fused step does not otherwise use: "loss")
        }
        if (dn && !mix) {
            hipStream_t ms = (hipStream_t)a->stream;
            HIP_TRY(zero_fill_async(ms, dn_shards, 4096), "zero shards");
            HIP_TRY(launch_depth_normal(ms, a->width, a->height, dnj.fx, dnj.fy, depth, mdepth, normal, dnj.weight, dnj.depth_ratio, dn_gd, dn_gm,
                                        dn_gn, dn_shards), "depth_normal launch");
        }
        const int rc = backward_impl(a->stream, a->P, a->D, a->M, R, a->background, a->width, a->height, xyz, shs, nullptr, alpha, scal, 1.0f,
                                     rotn, nullptr, a->viewmatrix, a->projmatrix, a->cam_pos, a->tan_fovx, a->tan_fovy, 0.0f, a->radii,
                                     normal, cg.last, cb.last, ci.last, dssim ? grad_img : nullptr, nullptr, nullptr, dn_gd, dn_gm, nullptr,
                                     dn_gn, a->workspace, a->dL_dmean2D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                     a->require_coord, a->require_depth, 0, dssim ? nullptr : a->gt, color, l1_scale, &f);
        if (rc < 0) return rc;
        if (!g_pending.active) return R;           // synchronous forward: R is already the true count
        const int Rt = igs_rast_forward_finish();
        if (Rt != IGS_RAST_E_RETRY) return Rt;     // the true count, or an error
        // a tile overflowed its slab: the guarded update kernel has done nothing; go again with the enlarged slabs
    }
    return fail(IGS_RAST_E_INVALID, "igs_refine_step: internal retry failure");
}

// Morton order of the Gaussians' positions (sort.hip): perm[i] = index of the Gaussian that comes i-th along the Z-order curve of
// xyz quantised to `bits` bits per axis inside the box lohi = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z} (device memory).
extern "C" size_t igs_morton_order_scratch_bytes(int P) { return morton_scratch_bytes(P) + 256; }
extern "C" int igs_morton_order(void* stream, int P, const float* xyz, const float* lohi, int bits, void* scratch, int* perm)
{
    if (P < 0 || bits < 1 || bits > 10) return fail(IGS_RAST_E_INVALID, "igs_morton_order: bad sizes (1..10 bits per axis)");
    if (P == 0) return 0;
    if (!xyz || !lohi || !scratch || !perm) return fail(IGS_RAST_E_INVALID, "igs_morton_order: NULL pointer");
    HIP_TRY(launch_morton_order((hipStream_t)stream, P, xyz, lohi, bits, scratch, perm), "morton order launch");
    return 0;
}

// Test support: the per-tile sort of the slab binning on caller-made slabs (sort.hip: launch_tile_sort).  tile_count[T] instances per
// tile (reset to zero by the launch), pairs[T * slab] = depth bits << 32 | Gaussian id, out: point_list[T * slab] (ids, sorted by the
// 64-bit key inside every tile's slab), ranges[2 T], stats[4] ([1] = largest tile that overflowed its slab).  Everything device memory.
extern "C" int igs_debug_tile_sort(void* stream, int T, uint32_t* tile_count, const unsigned long long* pairs, uint32_t* point_list,
                                   uint32_t* ranges, int slab, uint32_t* stats, int P)
{
    if (T <= 0 || slab <= 0 || P <= 0) return fail(IGS_RAST_E_INVALID, "igs_debug_tile_sort: bad sizes");
    if (!tile_count || !pairs || !point_list || !ranges || !stats) return fail(IGS_RAST_E_INVALID, "igs_debug_tile_sort: NULL pointer");
    HIP_TRY(launch_tile_sort((hipStream_t)stream, (uint32_t)T, tile_count, (const uint64_t*)pairs, point_list, ranges, (uint32_t)slab, stats,
                             nullptr, (uint32_t)P), "tile_sort launch");
    return 0;
}

extern "C" int igs_rast_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                                     const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    if (P < 0) return fail(IGS_RAST_E_INVALID, "igs_rast_mark_visible: bad size");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail(IGS_RAST_E_INVALID, "igs_rast_mark_visible: NULL pointer");
    HIP_TRY(launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present), "mark_visible launch");
    return 0;
}

extern "C" int igs_rast_debug_dump(void* stream, int P, int R, int width, int height, const char* geom_buffer,
                                   const char* binning_buffer, const char* image_buffer, float* rec32, uint32_t* tiles,
                                   uint32_t* point_list, uint32_t* ranges, uint32_t* n_contrib)
{
    hipStream_t s = (hipStream_t)stream;
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    const size_t Tn = (size_t)gx * gy, HW = (size_t)width * height;
    const GeomLayout GL(P); const ImgLayout IL(HW, Tn); const BinLayout BL(R);
    if (geom_buffer && P > 0) {
        const char* g = align_ptr(geom_buffer);
        if (rec32) HIP_TRY(hipMemcpyAsync(rec32, g + GL.rec, (size_t)P * REC_F * 4, hipMemcpyDeviceToDevice, s), "dump rec");
        if (tiles) HIP_TRY(hipMemcpyAsync(tiles, g + GL.tiles, (size_t)P * 4, hipMemcpyDeviceToDevice, s), "dump tiles");
    }
    if (image_buffer) {
        const char* i = align_ptr(image_buffer);
        // per-tile lists are gathered into the reference's compact layout (identical already on the global-sort path)
        if (ranges)
            HIP_TRY(launch_compact_lists(s, (uint32_t)Tn, (const uint32_t*)(i + IL.ranges),
                                         binning_buffer ? (const uint32_t*)(align_ptr(binning_buffer) + BL.point_list) : nullptr, ranges,
                                         point_list, (uint32_t)(R > 0 ? R : 0)), "dump lists");
        if (n_contrib) HIP_TRY(hipMemcpyAsync(n_contrib, i + IL.n_contrib, HW * 8, hipMemcpyDeviceToDevice, s), "dump n_contrib");
    }
    return 0;
}

// Test hook: fills the LDS of every CU with signalling garbage (NaN bit patterns), so that a kernel which reads LDS it never wrote
// -- invisible on a fresh device, whose LDS reads as zeros -- shows up in small parity tests (round 2 found such a read only through
// the PSNR of a long run: a missing wait state left one moment of each wave's first row unwritten).
__global__ void __launch_bounds__(256) poison_lds_kernel(float* sink)
{
    extern __shared__ uint32_t lds_all[];
    for (int i = threadIdx.x; i < 16000; i += 256) lds_all[i] = 0x7FC0BEEFu + (uint32_t)i;
    __syncthreads();
    if (lds_all[(threadIdx.x * 7) % 16000] == 1u && sink) sink[0] = 1.f;      // (keeps the stores alive)
}
extern "C" int igs_rast_debug_poison_lds(void* stream)
{
    // 64000 bytes per workgroup: two workgroups per CU cover most of the 160 KB; 2048 workgroups = 8 per CU, so every CU's LDS
    // is swept several times whatever the placement
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(256), 64000, (hipStream_t)stream, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : IGS_RAST_E_HIP;
}

extern "C" int igs_sh_grad_from_view_colors(void* stream, int P, int D, int M, int n_views, const float* means3D, const float* campos,
                                            const float* color_grads, float clamp_grads, float* dL_dsh)
{
    if (P < 0 || M < 0 || M > 16 || D < 0 || D > 3 || n_views < 0 || n_views > IGS_MAX_EXCHANGE_VIEWS)
        return fail(IGS_RAST_E_INVALID, "igs_sh_grad_from_view_colors: bad sizes (at most 64 views)");
    if (P == 0 || M == 0) return 0;
    if (!means3D || !dL_dsh || (n_views > 0 && (!campos || !color_grads))) return fail(IGS_RAST_E_INVALID, "igs_sh_grad_from_view_colors: NULL pointer");
    HIP_TRY(launch_sh_grad_views((hipStream_t)stream, P, D, M, n_views, means3D, campos, color_grads, clamp_grads, dL_dsh), "sh_grad_views launch");
    return 0;
}

extern "C" int igs_adam_sh_from_view_colors(void* stream, int P, int D, int M, int n_views, const float* means3D, const float* campos,
                                            const float* color_grads, float clamp_grads, float* param_sh, float* exp_avg_sh, float* exp_avg_sq_sh,
                                            float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt)
{
    if (P < 0 || M < 0 || M > 16 || D < 0 || D > 3 || n_views < 0 || n_views > IGS_MAX_EXCHANGE_VIEWS)
        return fail(IGS_RAST_E_INVALID, "igs_adam_sh_from_view_colors: bad sizes (at most 64 views)");
    if (P == 0 || M == 0) return 0;
    if (!means3D || !param_sh || !exp_avg_sh || !exp_avg_sq_sh || (n_views > 0 && (!campos || !color_grads)))
        return fail(IGS_RAST_E_INVALID, "igs_adam_sh_from_view_colors: NULL pointer");
    HIP_TRY(launch_sh_adam_views((hipStream_t)stream, P, D, M, n_views, means3D, campos, color_grads, clamp_grads, param_sh, exp_avg_sh, exp_avg_sq_sh,
                                 lr / bias_correction1, beta1, beta2, eps, 1.0f / bias_correction2_sqrt), "sh_adam_views launch");
    return 0;
}

// The whole optimiser step of an N > 1 rank in ONE launch (after the exchange): dL/dSH rebuilt from the gathered per-view colour
// gradients and applied as in igs_adam_sh_from_view_colors, and the four small groups updated from their all-reduced gradients in
// `grad` -- same arithmetic as igs_adam_step_groups.  param / exp_avg / exp_avg_sq / grad are the flat buffers of igs_refine_step,
// off_* float offsets into them.  The directions use the positions as they are BEFORE this update (every thread reads its own
// Gaussian's position first).
extern "C" int igs_adam_exchange_step(void* stream, int P, int D, int M, int n_views, const float* campos, const float* color_grads,
                                      float clamp_grads, float* param, float* exp_avg, float* exp_avg_sq, const float* grad,
                                      size_t off_xyz, size_t off_rot, size_t off_sh, size_t off_opacity, size_t off_scale,
                                      float lr_xyz, float lr_rot, float lr_sh, float lr_opacity, float lr_scale,
                                      float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt)
{
    if (P < 0 || M < 0 || M > 16 || D < 0 || D > 3 || n_views < 0 || n_views > IGS_MAX_EXCHANGE_VIEWS)
        return fail(IGS_RAST_E_INVALID, "igs_adam_exchange_step: bad sizes (at most 64 views)");
    if (P == 0) return 0;
    if (!param || !exp_avg || !exp_avg_sq || !grad || (n_views > 0 && (!campos || !color_grads)))
        return fail(IGS_RAST_E_INVALID, "igs_adam_exchange_step: NULL pointer");
    SmallGroupsAdam sm;
    sm.param = param; sm.exp_avg = exp_avg; sm.exp_avg_sq = exp_avg_sq; sm.grad = grad;
    sm.off[0] = off_xyz; sm.off[1] = off_rot; sm.off[2] = off_opacity; sm.off[3] = off_scale;
    sm.lr_over_bc1[0] = lr_xyz / bias_correction1; sm.lr_over_bc1[1] = lr_rot / bias_correction1;
    sm.lr_over_bc1[2] = lr_opacity / bias_correction1; sm.lr_over_bc1[3] = lr_scale / bias_correction1;
    HIP_TRY(launch_sh_adam_views((hipStream_t)stream, P, D, M, n_views, param + off_xyz, campos, color_grads, clamp_grads, param + off_sh,
                                 exp_avg + off_sh, exp_avg_sq + off_sh, lr_sh / bias_correction1, beta1, beta2, eps, 1.0f / bias_correction2_sqrt,
                                 &sm), "adam_exchange_step launch");
    return 0;
}
