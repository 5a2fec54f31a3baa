// Context, error reporting, version.
#include "common.hpp"

namespace spsp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace spsp

extern "C" {

const char* spsp_last_error(void) { return spsp::g_err; }
const char* spsp_version(void) { return "spsp 0.1 (gfx950)"; }

int spsp_init(int device, spsp_ctx** out) {
    SPSP_REQUIRE(out != nullptr, "spsp_init: out is NULL");
    int n = 0;
    SPSP_HIP_CHECK(hipGetDeviceCount(&n));
    SPSP_REQUIRE(device >= 0 && device < n, "spsp_init: device %d out of range (%d visible)", device, n);
    SPSP_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    SPSP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    spsp_ctx* c = new spsp_ctx;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    *out = c;
    return SPSP_OK;
}

void spsp_destroy(spsp_ctx* ctx) {
    if (!ctx) return;
    for (auto& r : ctx->prof_recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    for (auto& e : ctx->prof_free) (void)hipEventDestroy(e);
    delete ctx;
}

// One row per profiled kernel id (common.hpp, KernelId): the name bench.py / profiles/collect_traffic.py match against
// rocprofv3's kernel names (template arguments after the ones listed are element type / buffering flags), the
// algorithmic FLOPs ONE launch does per crop when every launch of the kernel does the same work (0 otherwise: not a
// roofline candidate), and which launch plans of a batch of 32 launch it: bit SPSP_PLAN_THROUGHPUT (what PlayerPipeline gives its
// slots; layer3 on conv_p4) | bit SPSP_PLAN_LATENCY (a fresh handle's plan; layer3 on conv_patch<false>).
struct KernelRow { const char* name; double flop_per_crop; int plans; };
static const int kTP = 1 << SPSP_PLAN_THROUGHPUT, kLT = 1 << SPSP_PLAN_LATENCY, kBoth = kTP | kLT;
static const double kFlopConv1 = 2.0 * 924844032.0, kFlop3x3 = 2.0 * 150994944.0;
static const double kFlopS2 = 2.0 * (75497472.0 + 8388608.0);     // a 3x3 / 2 conv + its 1x1 / 2 downsample
static const KernelRow kKernels[spsp::kKNumKernels] = {
    {"conv1_pool_kernel", kFlopConv1, kBoth}, {"conv3x3_wres_kernel", kFlop3x3, 0},
    {"conv3x3_patch_kernel<false>", kFlop3x3, kLT}, {"conv3x3_patch_kernel<true>", kFlop3x3, kBoth}, {"conv_ring_kernel", 0, 0},
    {"ief_fused_kernel", 0, kBoth}, {"ief_init_only_kernel", 0, 0}, {"proxy_pack_kernel", 0, 0}, {"proxy_build_kernel", 0, 0},
    {"rot6d_kernel", 0, 0}, {"fit_kernel2", 0, kBoth}, {"smpl_chain_kernel", 0, 0}, {"smpl_skin_kernel", 0, kBoth},
    {"smpl_extra_joints_kernel", 0, kBoth}, {"relate_project_kernel+relate_attend_kernel", 0, 0}, {"conv3x3s2_kernel", kFlopS2, kBoth}, {"conv3x3_p4_kernel", kFlop3x3, kBoth},
    {"conv3x3s2q_kernel", kFlopS2, kBoth}, {"conv3x3rs_kernel", kFlop3x3, kTP}, {"conv_basic_block_kernel", 2.0 * kFlop3x3, kBoth}};

int spsp_profile_num_kernels(void) { return spsp::kKNumKernels; }
const char* spsp_profile_kernel_name(int id) { return (id >= 0 && id < spsp::kKNumKernels) ? kKernels[id].name : nullptr; }
double spsp_profile_kernel_flop_per_crop(int id) { return (id >= 0 && id < spsp::kKNumKernels) ? kKernels[id].flop_per_crop : 0.0; }
int spsp_profile_kernel_in_plan(int id, int plan) {
    return (id >= 0 && id < spsp::kKNumKernels && (plan == SPSP_PLAN_THROUGHPUT || plan == SPSP_PLAN_LATENCY)) ? (kKernels[id].plans >> plan) & 1 : 0;
}

int spsp_profile_enable(spsp_ctx* ctx, int on) {
    SPSP_REQUIRE(ctx, "spsp_profile_enable: NULL ctx");
    ctx->prof_on = on != 0;
    return SPSP_OK;
}

int spsp_profile_select(spsp_ctx* ctx, int kernel_id) {
    SPSP_REQUIRE(ctx && kernel_id >= -1 && kernel_id < spsp::kKNumKernels, "spsp_profile_select: kernel id out of range");
    ctx->prof_only = kernel_id;
    return SPSP_OK;
}

int spsp_profile_timeline(spsp_ctx* ctx, int max_records, int* ids, unsigned long long* streams, double* start_ms,
                          double* stop_ms, int* count) {
    SPSP_REQUIRE(ctx && ids && streams && start_ms && stop_ms && count && max_records >= 0, "spsp_profile_timeline: bad arguments");
    int n = 0;
    for (auto& r : ctx->prof_recs) {
        SPSP_HIP_CHECK(hipEventSynchronize(r.stop));
        if (n < max_records) {
            float a = 0.f, b = 0.f;
            SPSP_HIP_CHECK(hipEventElapsedTime(&a, ctx->prof_recs.front().start, r.start));
            SPSP_HIP_CHECK(hipEventElapsedTime(&b, ctx->prof_recs.front().start, r.stop));
            ids[n] = r.id; streams[n] = (unsigned long long)(uintptr_t)r.stream; start_ms[n] = a; stop_ms[n] = b;
            ++n;
        }
        ctx->prof_free.push_back(r.start);
        ctx->prof_free.push_back(r.stop);
    }
    ctx->prof_recs.clear();
    *count = n;
    return SPSP_OK;
}

int spsp_profile_read(spsp_ctx* ctx, int n, double* total_ms, int* launches) { return spsp_profile_read_grids(ctx, n, total_ms, launches, nullptr); }

int spsp_profile_read_grids(spsp_ctx* ctx, int n, double* total_ms, int* launches, double* total_workgroups) {
    SPSP_REQUIRE(ctx && total_ms && launches && n >= spsp::kKNumKernels, "spsp_profile_read: need %d slots", spsp::kKNumKernels);
    for (int i = 0; i < n; ++i) { total_ms[i] = 0.0; launches[i] = 0; if (total_workgroups) total_workgroups[i] = 0.0; }
    for (auto& r : ctx->prof_recs) {
        SPSP_HIP_CHECK(hipEventSynchronize(r.stop));
        float ms = 0.f;
        SPSP_HIP_CHECK(hipEventElapsedTime(&ms, r.start, r.stop));
        total_ms[r.id] += ms;
        launches[r.id] += 1;
        if (total_workgroups) total_workgroups[r.id] += r.workgroups;
        ctx->prof_free.push_back(r.start);
        ctx->prof_free.push_back(r.stop);
    }
    ctx->prof_recs.clear();
    return SPSP_OK;
}

}  // extern "C"
