// torch_ops.cpp — libtorch host side of the MI355X rasterizer: three autograd Functions with
// OpenSplat's operator signatures, calling the C ABI of libgsplat_hip.so (include/gsplat_hip.h).
//
// Mirrors, on the reference side: project_gaussians.cpp:5-90, rasterize_gaussians.cpp:6-140,
// spherical_harmonics.cpp:3-62 and the allocation/launch glue of rasterizer/gsplat/bindings.cu.
// PyTorch is plumbing here (caching allocator, current HIP stream, autograd graph); no tensor
// math of the hot path runs in ATen.
//
// There is deliberately NO CPU fallback: every operator TORCH_CHECKs that its inputs live on the
// GPU, and the package fails to import if libgsplat_hip.so is missing.
#include "gsplat_ops.hpp"
#include "bindings_hip_native.h"

#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/gsplat_hip.h"
#include "../../include/gsplat_train.h"
#include "../../include/gsplat_densify.h"
#include "../../include/gsplat_dist.h"

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;
using torch::autograd::variable_list;

namespace {

std::atomic<bool> g_fast_exp{false};
std::atomic<bool> g_segmented{true};   // SplatRender: checkpointed forward + segmented backward on frames of few tiles

#define GS_CHECK_DEV(x) TORCH_CHECK((x).is_cuda(), #x " must be a GPU (HIP) tensor")
#define GS_CHECK_F32(x) TORCH_CHECK((x).scalar_type() == torch::kFloat32, #x " must be float32")
#define GS_CHECK_I32(x) TORCH_CHECK((x).scalar_type() == torch::kInt32, #x " must be int32")

void check_status(int rc, const char *what) {
    TORCH_CHECK(rc == GS_OK, what, " failed: ", gs_strerror(rc),
                rc == GS_ERR_HIP ? std::string(" — ") + gs_last_hip_error() : std::string());
}

// The stream argument of EVERY C-ABI call of this library is fetched here (bindings_hip_native.cpp: stream()),
// i.e. before the call is made: the place where the ABI check runs by itself, once per process, for every
// consumer — OpenSplat patched by integration/apply_hip_native.py included — without throwing inside dlopen
// (a static initialiser that throws ends in std::terminate; ADVICE r05: a check only ops.py calls
// protects nobody else).
static std::once_flag g_abiOnce;
gs_stream_t current_stream() {
    std::call_once(g_abiOnce, [] { gsplatCheckAbi(); });   // (an exception leaves the flag unset: checked again)
    return (gs_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
}

const float *fptr(const Tensor &t) { return t.data_ptr<float>(); }
float *fptr_mut(Tensor &t) { return t.data_ptr<float>(); }

GsCamera make_camera(double fx, double fy, double cx, double cy, int64_t H, int64_t W, double clip,
                     double glob, uint32_t flags = 0) {
    GsCamera c;
    std::memset(&c, 0, sizeof(c));
    c.flags = flags;
    c.fx = (float)fx; c.fy = (float)fy; c.cx = (float)cx; c.cy = (float)cy;
    c.img_width = (int32_t)W; c.img_height = (int32_t)H;
    c.clip_thresh = (float)clip; c.glob_scale = (float)glob;
    return c;
}

// 4x4 matrix argument: device tensors are handed to the kernel as they are (no sync); host
// tensors are copied into the camera struct.
const float *matrix_arg(const Tensor &m, Tensor &holder, float *host_dst) {
    TORCH_CHECK(m.numel() == 16, "view/projection matrix must be 4x4");
    holder = m.to(torch::kFloat32).contiguous();
    if (holder.is_cuda()) return holder.data_ptr<float>();
    std::memcpy(host_dst, holder.data_ptr<float>(), 16 * sizeof(float));
    return nullptr;
}

// float[3] arguments (background, camera position): the kernels read host or device memory, so a
// device tensor is handed over as it is — no copy to the host, no synchronisation
const float *vec3_arg(const Tensor &t, Tensor &holder) {
    holder = t.detach().to(torch::kFloat32).contiguous();
    return holder.data_ptr<float>();
}


// ---- the cov2d side channel of the reference-signature call ---------------------------------------
// An unmodified Model::forward (model.cpp:208-218) hands RasterizeGaussians::apply the reference's TEN
// arguments: no cov2d.  The exact gsplat-cpu pixel rectangle (gsplat_cpu.cpp:167-168) needs the
// projection's own cov2d — re-deriving it as conic^-1 moves a floor / ceil edge whenever the det clamp
// bound or the fp32 inversion rounds the other way ("what's weak" 1).  So
// ProjectGaussians::forward puts conics and cov2d into ONE storage ([2, N, 3] floats: conics first) and
// remembers that storage here; RasterizeGaussians::forward, given a `conics` tensor whose storage is one
// of the remembered ones (same StorageImpl object — the registry holds weak references, so a live entry
// proves identity, not just an equal address), unmodified since (version counter), finds the frame's
// cov2d right behind it.  Anything else (a clone, an edited or hand-made conics tensor) misses and takes
// the inversion; misses are counted (gsplatCov2dChannelCounters).
struct Cov2dChannelEntry {
    c10::weak_intrusive_ptr<c10::StorageImpl> storage;
    int64_t n;
    uint32_t version;
};
std::mutex g_cov2dMutex;
std::vector<Cov2dChannelEntry> g_cov2dChannel;
std::atomic<int64_t> g_cov2dHits{0}, g_cov2dMisses{0};

void cov2d_channel_register(const Tensor &conics, int64_t N) {
    std::lock_guard<std::mutex> lock(g_cov2dMutex);
    // entries whose storage is gone (the frame's tensors were released) leave
    g_cov2dChannel.erase(std::remove_if(g_cov2dChannel.begin(), g_cov2dChannel.end(),
                                        [](const Cov2dChannelEntry &e) { return e.storage.expired(); }),
                         g_cov2dChannel.end());
    g_cov2dChannel.push_back({c10::weak_intrusive_ptr<c10::StorageImpl>(conics.storage().getWeakStorageImpl()),
                              N, (uint32_t)conics._version()});
}

// the cov2d that ProjectGaussians::forward left behind `conics`, or an undefined tensor
Tensor cov2d_channel_lookup(const Tensor &conics, int64_t N) {
    if (!conics.defined() || !conics.is_contiguous() || conics.storage_offset() != 0 ||
        conics.numel() != 3 * N || N == 0)
        return Tensor();
    const c10::Storage &st = conics.storage();
    if (st.nbytes() < (size_t)(6 * N) * sizeof(float)) return Tensor();
    const c10::StorageImpl *impl = st.unsafeGetStorageImpl();
    bool found = false;
    {
        std::lock_guard<std::mutex> lock(g_cov2dMutex);
        for (const Cov2dChannelEntry &e : g_cov2dChannel)
            if (!e.storage.expired() && e.storage._unsafe_get_target() == impl && e.n == N &&
                e.version == (uint32_t)conics._version()) {
                found = true;
                break;
            }
    }
    if (!found) return Tensor();
    // a plain tensor over the second half of the storage (not an autograd view of `conics`)
    Tensor cov2d = torch::empty({0}, conics.options().requires_grad(false));
    cov2d.set_(st, 3 * N, {N, 3}, {3, 1});
    return cov2d;
}

}  // namespace

void gsplatSetFastExp(bool enabled) { g_fast_exp.store(enabled); }
bool gsplatGetFastExp() { return g_fast_exp.load(); }
void gsplatSetSegmentedBackward(bool enabled) { g_segmented.store(enabled); }
bool gsplatGetSegmentedBackward() { return g_segmented.load(); }

// ---- spherical_harmonics.cpp:3-28 helpers ------------------------------------------------------
int degFromSh(int numBases) {
    switch (numBases) {
    case 1: return 0;
    case 4: return 1;
    case 9: return 2;
    case 16: return 3;
    default: return 4;
    }
}

static const double kShC0 = 0.28209479177387814;
Tensor rgb2sh(const Tensor &rgb) { return (rgb - 0.5) / kShC0; }
Tensor sh2rgb(const Tensor &sh) { return torch::clamp((sh * kShC0) + 0.5, 0.0f, 1.0f); }

// ---- ProjectGaussians ---------------------------------------------------------------------------
variable_list ProjectGaussians::forward(AutogradContext *ctx, Tensor means, Tensor scales,
                                        double globScale, Tensor quats, Tensor viewMat,
                                        Tensor projMat, double fx, double fy, double cx, double cy,
                                        int64_t imgHeight, int64_t imgWidth, TileBounds tileBounds,
                                        double clipThresh) {
    (void)tileBounds;  // derived from imgWidth/imgHeight, as rasterize_gaussians.cpp:54-58 does
    GS_CHECK_DEV(means); GS_CHECK_DEV(scales); GS_CHECK_DEV(quats);
    GS_CHECK_F32(means); GS_CHECK_F32(scales); GS_CHECK_F32(quats);
    TORCH_CHECK(means.dim() == 2 && means.size(1) == 3, "means must be [N,3]");
    const int64_t N = means.size(0);
    TORCH_CHECK(scales.sizes() == means.sizes(), "scales must be [N,3]");
    TORCH_CHECK(quats.dim() == 2 && quats.size(0) == N && quats.size(1) == 4, "quats must be [N,4]");
    c10::DeviceGuard guard(means.device());
    means = means.contiguous(); scales = scales.contiguous(); quats = quats.contiguous();

    GsCamera cam = make_camera(fx, fy, cx, cy, imgHeight, imgWidth, clipThresh, globScale);
    Tensor vmHold, pmHold;
    const float *vmDev = matrix_arg(viewMat, vmHold, cam.viewmat);
    const float *pmDev = matrix_arg(projMat, pmHold, cam.projmat);

    auto f32 = means.options();
    auto i32 = means.options().dtype(torch::kInt32);
    Tensor xys = torch::empty({N, 2}, f32), depths = torch::empty({N}, f32);
    Tensor radii = torch::empty({N}, i32);
    Tensor numTilesHit = torch::empty({N}, i32), cov3d = torch::empty({N, 6}, f32);
    // conics and cov2d share ONE storage (conics first): the reference-signature rasterize call, which
    // receives only `conics`, finds the frame's cov2d behind it (cov2d_channel_*, above).  Both are plain
    // tensors over that storage, not autograd views of each other.
    Tensor cc = torch::empty({2, N, 3}, f32);
    Tensor conics = torch::empty({0}, f32), cov2d = torch::empty({0}, f32);
    conics.set_(cc.storage(), 0, {N, 3}, {3, 1});
    cov2d.set_(cc.storage(), 3 * N, {N, 3}, {3, 1});
    check_status(gs_project_forward(&cam, vmDev, pmDev, (int)N, fptr(means), fptr(scales),
                                    fptr(quats), fptr_mut(xys), fptr_mut(depths),
                                    radii.data_ptr<int32_t>(), fptr_mut(conics),
                                    numTilesHit.data_ptr<int32_t>(), fptr_mut(cov3d),
                                    fptr_mut(cov2d), current_stream()),
                 "gs_project_forward");

    ctx->saved_data["imgHeight"] = imgHeight;
    ctx->saved_data["imgWidth"] = imgWidth;
    ctx->saved_data["globScale"] = globScale;
    ctx->saved_data["fx"] = fx; ctx->saved_data["fy"] = fy;
    ctx->saved_data["cx"] = cx; ctx->saved_data["cy"] = cy;
    ctx->saved_data["clipThresh"] = clipThresh;
    ctx->save_for_backward({means, scales, quats, vmHold, pmHold, radii});
    ctx->mark_non_differentiable({radii, numTilesHit, cov3d, cov2d});
    cov2d_channel_register(conics, N);
    return {xys, depths, radii, conics, numTilesHit, cov3d, cov2d};
}

tensor_list ProjectGaussians::backward(AutogradContext *ctx, tensor_list grad_outputs) {
    variable_list saved = ctx->get_saved_variables();
    Tensor means = saved[0], scales = saved[1], quats = saved[2];
    Tensor viewMat = saved[3], projMat = saved[4], radii = saved[5];
    const int64_t N = means.size(0);
    c10::DeviceGuard guard(means.device());

    Tensor v_xys = grad_outputs[0].defined() ? grad_outputs[0].contiguous()
                                             : torch::zeros({N, 2}, means.options());
    Tensor v_depths = grad_outputs[1].defined() ? grad_outputs[1].contiguous() : Tensor();
    Tensor v_conics = grad_outputs[3].defined() ? grad_outputs[3].contiguous()
                                                : torch::zeros({N, 3}, means.options());
    GsCamera cam = make_camera(ctx->saved_data["fx"].toDouble(), ctx->saved_data["fy"].toDouble(),
                               ctx->saved_data["cx"].toDouble(), ctx->saved_data["cy"].toDouble(),
                               ctx->saved_data["imgHeight"].toInt(),
                               ctx->saved_data["imgWidth"].toInt(),
                               ctx->saved_data["clipThresh"].toDouble(),
                               ctx->saved_data["globScale"].toDouble());
    Tensor vmHold, pmHold;
    const float *vmDev = matrix_arg(viewMat, vmHold, cam.viewmat);
    const float *pmDev = matrix_arg(projMat, pmHold, cam.projmat);

    Tensor v_means = torch::empty({N, 3}, means.options());
    Tensor v_scales = torch::empty({N, 3}, means.options());
    Tensor v_quats = torch::empty({N, 4}, means.options());
    check_status(gs_project_backward(&cam, vmDev, pmDev, (int)N, fptr(means), fptr(scales),
                                     fptr(quats), radii.data_ptr<int32_t>(), fptr(v_xys),
                                     v_depths.defined() ? fptr(v_depths) : nullptr, fptr(v_conics),
                                     fptr_mut(v_means), fptr_mut(v_scales), fptr_mut(v_quats),
                                     current_stream()),
                 "gs_project_backward");
    Tensor none;
    return {v_means, v_scales, none, v_quats, none, none, none, none, none, none, none, none, none,
            none};
}

// ---- binning ------------------------------------------------------------------------------------
// State of the speculative binning, one record per (device, image width, image height): training
// renders at 1/4, 1/2 and full resolution (model.cpp:249-251) and validates at full resolution, and
// each of those sees its own intersection counts.
//   capacity   entries the id list is given without asking the device: the RUNNING MAXIMUM of
//              1.125 M + 1024 over every frame validated so far.  It never shrinks by itself
//              (gsplatResetBinningState does that): with OpenSplat's random camera order a hint that
//              followed the LAST frame would repeat the forward for every camera whose M exceeds its
//              predecessor's by more than 12.5 %;
//   listStats  {M, longest tile list} of the last validated frame of this key: only a hint for the
//              binning of the NEXT frame (which size classes of the per-tile sort to launch).  The
//              compositing launches of a frame use that frame's OWN statistics, which travel with its
//              BinnedLists and, for the backward, in the autograd node's saved_data.
// The reference blocks on cumsum().item() instead (rasterize_gaussians.cpp:62-63).
struct BinState {
    int64_t capacity = 0;
    int32_t listStats[2] = {0, 0};
};
static std::mutex g_binMutex;
static std::map<std::tuple<int, int, int>, BinState> g_binStates;
static std::atomic<int64_t> g_binCalls{0}, g_binRepeats{0};

static BinState readBinState(int device, int W, int H) {
    std::lock_guard<std::mutex> lock(g_binMutex);
    return g_binStates[std::make_tuple(device, W, H)];
}

// Binning of already-packed records (gs_pack_splats or gs_gaussian_forward): count + scan, scatter,
// per-tile sort, coverage masks — all enqueued, nothing waited for.
BinnedLists binPackedRecords(const Tensor &packed, const Tensor &depths, int H, int W) {
    const int64_t N = depths.size(0);
    auto i32 = depths.options().dtype(torch::kInt32);
    gs_stream_t s = current_stream();
    BinnedLists b;
    b.packed = packed;
    b.width = W; b.height = H;
    b.device = depths.get_device();
    const BinState st = readBinState(b.device, W, H);
    // The id list is sized from the intersection counts this process has seen for this image size and
    // the count of THIS call is only read back after the caller has enqueued the compositing kernel
    // (validateBinning): the stream never idles waiting for the host.  A guess that was too small
    // costs one repeat.
    const int tiles = ((W + GS_TILE - 1) / GS_TILE) * ((H + GS_TILE - 1) / GS_TILE);
    b.tileBins = torch::empty({tiles, 2}, i32);
    b.count = torch::zeros({2}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
    const int64_t cap = std::max<int64_t>(st.capacity, 1024);
    b.gaussianIdsSorted = torch::empty({cap}, i32);
    b.blockMasks = torch::empty({cap}, depths.options().dtype(torch::kInt16));
    size_t wsBytes = gs_bin_workspace_bytes((int)N, cap, W, H);
    Tensor ws = torch::empty({(int64_t)wsBytes}, depths.options().dtype(torch::kUInt8));
    // tiles by descending list length: the compositing launches start with the long lists
    b.tileOrder = torch::empty({tiles}, i32);
    // (one call: count, scatter with the scan folded in — tile_bins, tile order and the counts come from an extra
    // workgroup of the scatter launch —, per-tile sorts)
    check_status(gs_bin_speculative(W, H, (int)N, (int32_t)cap, fptr(packed), fptr(depths),
                                    b.tileBins.data_ptr<int32_t>(), b.gaussianIdsSorted.data_ptr<int32_t>(),
                                    reinterpret_cast<uint16_t *>(b.blockMasks.data_ptr<int16_t>()),
                                    b.tileOrder.data_ptr<int32_t>(), b.count.data_ptr<int32_t>(), st.listStats,
                                    ws.data_ptr(), wsBytes, s),
                 "gs_bin_speculative");
    // validateBinning waits for this event (the counts are in pinned memory), not for the stream
    auto scanDone = std::make_shared<at::cuda::CUDAEvent>();
    scanDone->record(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
    b.scanDone = scanDone;
    g_binCalls++;
    return b;
}

// rasterize_gaussians.cpp:6-37, statement for statement on this library's launchers
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor>
binAndSortGaussians(int numPoints, int numIntersects, Tensor xys, Tensor depths, Tensor radii,
                    Tensor cumTilesHit, TileBounds tileBounds) {
    auto t = map_gaussian_to_intersects_tensor(numPoints, numIntersects, xys, depths, radii, cumTilesHit,
                                               tileBounds);
    Tensor isectIds = std::get<0>(t), gaussianIds = std::get<1>(t);
    auto sorted = torch::sort(isectIds);
    Tensor isectIdsSorted = std::get<0>(sorted), sortedIndices = std::get<1>(sorted);
    Tensor gaussianIdsSorted = torch::gather(gaussianIds, 0, sortedIndices);
    Tensor tileBins = get_tile_bin_edges_tensor(numIntersects, isectIdsSorted);
    return std::make_tuple(isectIds, gaussianIds, isectIdsSorted, gaussianIdsSorted, tileBins);
}

BinnedLists binAndSortPacked(const Tensor &xys, const Tensor &depths, const Tensor &radii,
                             const Tensor &conics, const Tensor &colors, const Tensor &opacity,
                             const Tensor &cov2d, int imgHeight, int imgWidth, bool opacityIsLogit) {
    const int64_t N = xys.size(0);
    const int W = imgWidth, H = imgHeight;
    auto f32 = xys.options().dtype(torch::kFloat32);
    auto i32 = xys.options().dtype(torch::kInt32);
    gs_stream_t s = current_stream();

    Tensor packed = torch::empty({N, GS_SPLAT_DWORDS}, f32);
    Tensor tilesHit = torch::empty({N}, i32);
    check_status(gs_pack_splats(W, H, (int)N, fptr(xys), radii.data_ptr<int32_t>(), fptr(conics),
                                fptr(colors), fptr(opacity),
                                cov2d.defined() ? fptr(cov2d) : nullptr, fptr_mut(packed),
                                tilesHit.data_ptr<int32_t>(),
                                opacityIsLogit ? GS_FLAG_LOGIT_OPACITY : 0u, s),
                 "gs_pack_splats");

    return binPackedRecords(packed, depths, H, W);
}

// Waits until the scan kernel of `b` has stored its intersection count in pinned memory (an event
// wait: kernels enqueued behind it keep running, the host goes on enqueuing afterwards), records the
// frame's statistics in `b` and in the per-(device, size) state, and checks the count against the
// capacity the id list was given.  false -> the lists were truncated: repeat binning + compositing.
bool validateBinning(BinnedLists &b) {
    if (b.scanDone) std::static_pointer_cast<at::cuda::CUDAEvent>(b.scanDone)->synchronize();
    else c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().synchronize();
    const int64_t M = b.count.data_ptr<int32_t>()[0];
    b.listStats[0] = (int32_t)M;
    b.listStats[1] = b.count.data_ptr<int32_t>()[1];
    {
        std::lock_guard<std::mutex> lock(g_binMutex);
        BinState &st = g_binStates[std::make_tuple(b.device, b.width, b.height)];
        st.capacity = std::max<int64_t>(st.capacity, M + M / 8 + 1024);   // running maximum
        st.listStats[0] = b.listStats[0];
        st.listStats[1] = b.listStats[1];
    }
    const bool ok = M <= b.gaussianIdsSorted.size(0);
    if (!ok) g_binRepeats++;
    return ok;
}

// The checkpoint buffer of one forward / backward pair of the compositing kernels, planned from the statistics of
// the last validated frame of this (device, size) — gsplat_hip.h: gs_rasterize_checkpoint_plan; bytes = 0: the
// plain schedule (a full frame, short lists, no statistics yet, or gsplatSetSegmentedBackward(false)).
struct CheckpointPlan {
    Tensor buffer;
    size_t bytes = 0;
    int32_t segLen = 0, maxSegments = 0;
    void *ptr() const { return bytes ? buffer.data_ptr() : nullptr; }
    Tensor saved(const torch::TensorOptions &f32) const {
        return bytes ? buffer : torch::empty({0}, f32.dtype(torch::kUInt8));
    }
};
// needGrad: a forward none of whose inputs requires a gradient (evaluation renders) has no backward: no plan, no
// records written
static CheckpointPlan planCheckpoints(int device, int W, int H, const torch::TensorOptions &f32, bool needGrad) {
    CheckpointPlan cp;
    if (!g_segmented.load() || !needGrad) return cp;
    const BinState st = readBinState(device, W, H);
    check_status(gs_rasterize_checkpoint_plan(W, H, st.listStats, &cp.segLen, &cp.maxSegments, &cp.bytes),
                 "gs_rasterize_checkpoint_plan");
    // (sizes follow the previous frame's longest list: rounded up to 32 MiB steps, so that the caching allocator
    // re-uses a handful of block sizes instead of collecting one per frame)
    if (cp.bytes) {
        const int64_t step = 32ll << 20;
        cp.buffer = torch::empty({((int64_t)cp.bytes + step - 1) / step * step}, f32.dtype(torch::kUInt8));
    }
    return cp;
}

std::tuple<int64_t, int64_t> gsplatCov2dChannelCounters(bool reset) {
    auto r = std::make_tuple(g_cov2dHits.load(), g_cov2dMisses.load());
    if (reset) { g_cov2dHits = 0; g_cov2dMisses = 0; }
    return r;
}

void gsplatResetBinningState() {
    std::lock_guard<std::mutex> lock(g_binMutex);
    g_binStates.clear();
    g_binCalls = 0;
    g_binRepeats = 0;
}

std::tuple<int64_t, int64_t> gsplatBinningCounters() {
    return std::make_tuple(g_binCalls.load(), g_binRepeats.load());
}

int64_t gsplatBinningCapacity(int device, int imgWidth, int imgHeight) {
    return readBinState(device, imgWidth, imgHeight).capacity;
}

static const uint16_t *maskptr(const Tensor &m) {
    return reinterpret_cast<const uint16_t *>(m.data_ptr<int16_t>());
}

// ---- RasterizeGaussians -------------------------------------------------------------------------
Tensor RasterizeGaussians::forward(AutogradContext *ctx, Tensor xys, Tensor depths, Tensor radii,
                                   Tensor conics, Tensor numTilesHit, Tensor colors,
                                   Tensor opacity, int64_t imgHeight, int64_t imgWidth,
                                   Tensor background, c10::optional<Tensor> cov2dOpt) {
    Tensor cov2d = (cov2dOpt.has_value() && cov2dOpt->defined()) ? *cov2dOpt : Tensor();
    // numTilesHit carries the reference's radius-square counts (forward.cu:86-94), which size ITS global
    // key list (rasterize_gaussians.cpp:62-63).  The lists here are per tile of the gsplat-cpu rectangle
    //  and are counted on the device: the argument is validated and not read.
    TORCH_CHECK(!numTilesHit.defined() || numTilesHit.numel() == xys.size(0),
                "numTilesHit must have N elements");
    GS_CHECK_DEV(xys); GS_CHECK_DEV(depths); GS_CHECK_DEV(radii); GS_CHECK_DEV(conics);
    GS_CHECK_DEV(colors); GS_CHECK_DEV(opacity);
    GS_CHECK_F32(xys); GS_CHECK_F32(depths); GS_CHECK_I32(radii); GS_CHECK_F32(conics);
    GS_CHECK_F32(colors); GS_CHECK_F32(opacity);
    const int64_t N = xys.size(0);
    TORCH_CHECK(xys.dim() == 2 && xys.size(1) == 2, "xys must be [N,2]");
    TORCH_CHECK(conics.dim() == 2 && conics.size(0) == N && conics.size(1) == 3, "conics must be [N,3]");
    TORCH_CHECK(colors.dim() == 2 && colors.size(0) == N && colors.size(1) == 3,
                "colors must be [N,3] (3 channels; forward.cu:256-378)");
    TORCH_CHECK(opacity.numel() == N, "opacity must have N elements");
    TORCH_CHECK(background.numel() == 3, "background must have 3 elements");
    if (cov2d.defined()) {
        GS_CHECK_DEV(cov2d); GS_CHECK_F32(cov2d);
        TORCH_CHECK(cov2d.numel() == 3 * N, "cov2d must be [N,3] = (xx, xy, yy)");
        cov2d = cov2d.contiguous();
    }
    c10::DeviceGuard guard(xys.device());
    xys = xys.contiguous(); depths = depths.contiguous(); radii = radii.contiguous();
    if (!cov2d.defined()) {
        // the reference's ten-argument form (rasterize_gaussians.hpp:23-37): cov2d from the side channel
        // of ProjectGaussians::forward; only a `conics` that is not that operator's untouched output
        // falls back to conic^-1
        cov2d = cov2d_channel_lookup(conics, N);
        if (cov2d.defined()) {
            g_cov2dHits++;
        } else {
            g_cov2dMisses++;
            TORCH_WARN_ONCE("RasterizeGaussians (ten-argument form): `conics` is not the untouched output of "
                            "ProjectGaussians (a clone, an edited or a hand-made tensor), so the projection's "
                            "cov2d cannot be found behind it; the pixel rectangles are re-derived from conic^-1 "
                            "— close to, not identical with, rasterizer/gsplat-cpu.  Pass cov2d as the eleventh "
                            "argument, or hand over ProjectGaussians' conics as returned (.detach() / "
                            ".contiguous() of it are fine).  Further misses are only counted "
                            "(gsplatCov2dChannelCounters).");
        }
    }
    conics = conics.contiguous(); colors = colors.contiguous(); opacity = opacity.contiguous();
    const int W = (int)imgWidth, H = (int)imgHeight;

    uint32_t flags = g_fast_exp.load() ? GS_FLAG_FAST_EXP : 0u;
    Tensor bgHold;
    const float *bg = vec3_arg(background, bgHold);
    auto f32 = xys.options();
    Tensor outImg = torch::empty({H, W, 3}, f32), finalTs = torch::empty({H, W}, f32);
    Tensor finalIdx = torch::empty({H, W}, f32.dtype(torch::kInt32));
    // a frame that does not fill the chip, or with a tail of long lists (gsplat_hip.h:
    // gs_rasterize_checkpoint_plan; planned from the last frame of this size): checkpoints for the backward
    const bool needGrad = xys.requires_grad() || conics.requires_grad() || colors.requires_grad() ||
                          opacity.requires_grad();
    CheckpointPlan cp = planCheckpoints(xys.get_device(), W, H, f32, needGrad);
    BinnedLists b;
    for (;;) {
        b = binAndSortPacked(xys, depths, radii, conics, colors, opacity, cov2d, H, W, false);
        check_status(gs_rasterize_forward_ckpt(W, H, b.gaussianIdsSorted.data_ptr<int32_t>(),
                                               maskptr(b.blockMasks), b.tileBins.data_ptr<int32_t>(),
                                               fptr(b.packed), bg, fptr_mut(outImg), fptr_mut(finalTs),
                                               finalIdx.data_ptr<int32_t>(), nullptr, nullptr,
                                               b.tileOrder.numel() ? b.tileOrder.data_ptr<int32_t>() : nullptr,
                                               flags, cp.ptr(), cp.bytes, cp.segLen, cp.maxSegments,
                                               current_stream()),
                     "gs_rasterize_forward");
        if (validateBinning(b)) break;
    }
    Tensor packed = b.packed, idsSorted = b.gaussianIdsSorted, tileBins = b.tileBins, tileOrder = b.tileOrder;

    // this frame's own list statistics steer the backward's wave geometry
    ctx->saved_data["listM"] = (int64_t)b.listStats[0];
    ctx->saved_data["listLongest"] = (int64_t)b.listStats[1];
    ctx->saved_data["segLen"] = (int64_t)cp.segLen;
    ctx->saved_data["maxSegments"] = (int64_t)cp.maxSegments;
    ctx->saved_data["imgWidth"] = imgWidth;
    ctx->saved_data["imgHeight"] = imgHeight;
    ctx->saved_data["flags"] = (int64_t)flags;
    ctx->saved_data["numPoints"] = N;
    ctx->save_for_backward({idsSorted, tileBins, packed, finalTs, finalIdx, bgHold, tileOrder,
                            b.blockMasks, cp.saved(f32)});
    return outImg;
}

tensor_list RasterizeGaussians::backward(AutogradContext *ctx, tensor_list grad_outputs) {
    const int W = (int)ctx->saved_data["imgWidth"].toInt(), H = (int)ctx->saved_data["imgHeight"].toInt();
    const int64_t N = ctx->saved_data["numPoints"].toInt();
    variable_list saved = ctx->get_saved_variables();
    Tensor idsSorted = saved[0], tileBins = saved[1], packed = saved[2];
    Tensor finalTs = saved[3], finalIdx = saved[4], bgHold = saved[5], tileOrder = saved[6];
    Tensor blockMasks = saved[7], checkpoints = saved[8];
    const size_t ckBytes = (size_t)checkpoints.numel();
    const int32_t listStats[2] = {(int32_t)ctx->saved_data["listM"].toInt(),
                                  (int32_t)ctx->saved_data["listLongest"].toInt()};
    c10::DeviceGuard guard(packed.device());
    Tensor v_outImg = grad_outputs[0].contiguous();
    GS_CHECK_F32(v_outImg);
    const float *bg = bgHold.data_ptr<float>();
    auto f32 = packed.options();
    Tensor v_xy = torch::empty({N, 2}, f32), v_conic = torch::empty({N, 3}, f32);
    Tensor v_colors = torch::empty({N, 3}, f32), v_opacity = torch::empty({N, 1}, f32);
    const size_t wsBytes = gs_rasterize_backward_workspace_bytes((int)N);
    Tensor ws = torch::empty({(int64_t)(wsBytes ? wsBytes : 64)}, f32.dtype(torch::kUInt8));
    check_status(gs_rasterize_backward_ckpt(W, H, (int)N, idsSorted.data_ptr<int32_t>(),
                                            maskptr(blockMasks),
                                            tileBins.data_ptr<int32_t>(), fptr(packed), bg,
                                            fptr(finalTs), finalIdx.data_ptr<int32_t>(), fptr(v_outImg),
                                            nullptr /* v_out_alpha: zeros, rasterize_gaussians.cpp:108 */,
                                            nullptr, fptr_mut(v_xy), fptr_mut(v_conic), fptr_mut(v_colors),
                                            fptr_mut(v_opacity), ws.data_ptr(), wsBytes, listStats,
                                            tileOrder.numel() ? tileOrder.data_ptr<int32_t>() : nullptr,
                                            (uint32_t)ctx->saved_data["flags"].toInt(),
                                            ckBytes ? checkpoints.data_ptr() : nullptr, ckBytes,
                                            (int32_t)ctx->saved_data["segLen"].toInt(),
                                            (int32_t)ctx->saved_data["maxSegments"].toInt(), current_stream()),
                 "gs_rasterize_backward");
    Tensor none;
    return {v_xy, none, none, v_conic, none, v_colors, v_opacity, none, none, none, none};
}

// ---- SphericalHarmonics ---------------------------------------------------------------------------
Tensor SphericalHarmonics::forward(AutogradContext *ctx, int64_t degreesToUse, Tensor viewDirs,
                                   Tensor coeffs) {
    GS_CHECK_DEV(viewDirs); GS_CHECK_DEV(coeffs); GS_CHECK_F32(viewDirs); GS_CHECK_F32(coeffs);
    TORCH_CHECK(coeffs.dim() == 3 && coeffs.size(2) == 3, "coeffs must have dimensions (N, D, 3)");
    const int64_t N = coeffs.size(0), K = coeffs.size(1);
    TORCH_CHECK(viewDirs.dim() == 2 && viewDirs.size(0) == N && viewDirs.size(1) == 3,
                "viewDirs must be [N,3]");
    c10::DeviceGuard guard(coeffs.device());
    viewDirs = viewDirs.contiguous(); coeffs = coeffs.contiguous();
    Tensor colors = torch::empty({N, 3}, coeffs.options());
    check_status(gs_sh_forward((int)N, (int)K, (int)degreesToUse, fptr(viewDirs), fptr(coeffs),
                               fptr_mut(colors), current_stream()),
                 "gs_sh_forward");
    ctx->saved_data["degreesToUse"] = degreesToUse;
    ctx->saved_data["numBases"] = K;
    ctx->save_for_backward({viewDirs});
    return colors;
}

tensor_list SphericalHarmonics::backward(AutogradContext *ctx, tensor_list grad_outputs) {
    Tensor viewDirs = ctx->get_saved_variables()[0];
    const int64_t N = viewDirs.size(0), K = ctx->saved_data["numBases"].toInt();
    c10::DeviceGuard guard(viewDirs.device());
    Tensor v_colors = grad_outputs[0].contiguous();
    Tensor v_coeffs = torch::empty({N, K, 3}, viewDirs.options());
    check_status(gs_sh_backward((int)N, (int)K, (int)ctx->saved_data["degreesToUse"].toInt(),
                                fptr(viewDirs), fptr(v_colors), fptr_mut(v_coeffs), current_stream()),
                 "gs_sh_backward");
    Tensor none;
    return {none, none, v_coeffs};
}


// ---- SplatRender: Model::forward's render chain as ONE autograd node --------
// Replaces, with identical semantics, model.cpp:114 (cat), :147-159 (exp, normalise, project),
// :176-192 (view dirs, SH, +0.5, clamp_min), :208-218 (sigmoid, rasterize) and :222 (clamp_max):
// the element-wise glue runs inside the kernels (GS_CAM_LOG_SCALES, gs_sh_*_fused,
// GS_FLAG_LOGIT_OPACITY, GS_FLAG_CLAMP_IMAGE), no concatenated / exponentiated / normalised copies
// of the parameters are materialised.  xys is returned (detached) together with radii for
// Model::afterTrain (model.cpp:318-336); since xys is no longer a graph leaf whose .grad could be
// retained (model.cpp:171), its gradient is written into `xysGradOut` when that tensor is given.
variable_list SplatRender::forward(AutogradContext *ctx, Tensor means, Tensor logScales,
                                   Tensor quats, Tensor opacityLogits, Tensor featuresDc,
                                   Tensor featuresRest, Tensor viewMat, Tensor projMat,
                                   Tensor camPos, double fx, double fy, double cx, double cy,
                                   int64_t imgHeight, int64_t imgWidth, int64_t degreesToUse,
                                   Tensor background, c10::optional<Tensor> xysGradOut) {
    GS_CHECK_DEV(means); GS_CHECK_DEV(logScales); GS_CHECK_DEV(quats); GS_CHECK_DEV(opacityLogits);
    GS_CHECK_DEV(featuresDc);
    GS_CHECK_F32(means); GS_CHECK_F32(logScales); GS_CHECK_F32(quats); GS_CHECK_F32(opacityLogits);
    GS_CHECK_F32(featuresDc);
    const int64_t N = means.size(0);
    TORCH_CHECK(means.dim() == 2 && means.size(1) == 3, "means must be [N,3]");
    TORCH_CHECK(logScales.sizes() == means.sizes(), "scales must be [N,3]");
    TORCH_CHECK(quats.dim() == 2 && quats.size(0) == N && quats.size(1) == 4, "quats must be [N,4]");
    TORCH_CHECK(opacityLogits.numel() == N, "opacities must have N elements");
    TORCH_CHECK(featuresDc.dim() == 2 && featuresDc.size(0) == N && featuresDc.size(1) == 3,
                "featuresDc must be [N,3]");
    const bool hasRest = featuresRest.defined() && featuresRest.numel() > 0;
    if (hasRest) {
        GS_CHECK_DEV(featuresRest); GS_CHECK_F32(featuresRest);
        TORCH_CHECK(featuresRest.dim() == 3 && featuresRest.size(0) == N && featuresRest.size(2) == 3,
                    "featuresRest must be [N,K-1,3]");
    }
    const int64_t K = 1 + (hasRest ? featuresRest.size(1) : 0);
    TORCH_CHECK(background.numel() == 3, "background must have 3 elements");
    TORCH_CHECK(camPos.numel() == 3, "camPos must have 3 elements");
    c10::DeviceGuard guard(means.device());
    means = means.contiguous(); logScales = logScales.contiguous(); quats = quats.contiguous();
    opacityLogits = opacityLogits.contiguous(); featuresDc = featuresDc.contiguous();
    if (hasRest) featuresRest = featuresRest.contiguous();
    const int W = (int)imgWidth, H = (int)imgHeight;
    gs_stream_t s = current_stream();

    GsCamera cam = make_camera(fx, fy, cx, cy, imgHeight, imgWidth, 0.01, 1.0, GS_CAM_LOG_SCALES);
    Tensor vmHold, pmHold;
    const float *vmDev = matrix_arg(viewMat, vmHold, cam.viewmat);
    const float *pmDev = matrix_arg(projMat, pmHold, cam.projmat);
    Tensor cpHold;
    const float *cp = vec3_arg(camPos, cpHold);

    auto f32 = means.options();
    auto i32 = means.options().dtype(torch::kInt32);
    // projection + SH colour + packed record in one fused stage (gs_gaussian_forward)
    Tensor xys = torch::empty({N, 2}, f32), depths = torch::empty({N}, f32);
    Tensor radii = torch::empty({N}, i32), rgbRaw = torch::empty({N, 3}, f32);
    Tensor packedAll = torch::empty({N, GS_SPLAT_DWORDS}, f32);
    uint32_t flags = (g_fast_exp.load() ? GS_FLAG_FAST_EXP : 0u) | GS_FLAG_CLAMP_IMAGE |
                     GS_FLAG_LOGIT_OPACITY;
    check_status(gs_gaussian_forward(&cam, vmDev, pmDev, (int)N, (int)K, (int)degreesToUse, fptr(means),
                                     fptr(logScales), fptr(quats), fptr(opacityLogits),
                                     fptr(featuresDc), hasRest ? fptr(featuresRest) : nullptr, cp,
                                     fptr_mut(packedAll), fptr_mut(depths), radii.data_ptr<int32_t>(),
                                     fptr_mut(rgbRaw), fptr_mut(xys), flags, s),
                 "gs_gaussian_forward");
    Tensor imgRaw = torch::empty({H, W, 3}, f32), img = torch::empty({H, W, 3}, f32);
    Tensor finalTs = torch::empty({H, W}, f32), finalIdx = torch::empty({H, W}, i32);
    Tensor bgHold;
    const float *bg = vec3_arg(background, bgHold);
    // a frame of few tiles with long lists (the reduced resolutions a run starts with, model.cpp:85-92): the
    // forward leaves checkpoints along the lists, the backward runs their pieces side by side — planned
    // from the statistics of the last frame of this size (gsplat_hip.h: gs_rasterize_checkpoint_plan)
    const bool needGrad = means.requires_grad() || logScales.requires_grad() || quats.requires_grad() ||
                          opacityLogits.requires_grad() || featuresDc.requires_grad() ||
                          (hasRest && featuresRest.requires_grad());
    CheckpointPlan ckp = planCheckpoints(means.get_device(), W, H, f32, needGrad);
    BinnedLists b;
    for (;;) {
        b = binPackedRecords(packedAll, depths, H, W);
        check_status(gs_rasterize_forward_ckpt(W, H, b.gaussianIdsSorted.data_ptr<int32_t>(),
                                               maskptr(b.blockMasks), b.tileBins.data_ptr<int32_t>(),
                                               fptr(b.packed), bg, fptr_mut(imgRaw), fptr_mut(finalTs),
                                               finalIdx.data_ptr<int32_t>(), fptr_mut(img), nullptr,
                                               b.tileOrder.numel() ? b.tileOrder.data_ptr<int32_t>() : nullptr,
                                               flags, ckp.ptr(), ckp.bytes, ckp.segLen, ckp.maxSegments, s),
                     "gs_rasterize_forward");
        if (validateBinning(b)) break;
    }
    Tensor packed = b.packed, idsSorted = b.gaussianIdsSorted, tileBins = b.tileBins, tileOrder = b.tileOrder;
    ctx->saved_data["listM"] = (int64_t)b.listStats[0];
    ctx->saved_data["listLongest"] = (int64_t)b.listStats[1];
    ctx->saved_data["segLen"] = (int64_t)ckp.segLen;
    ctx->saved_data["maxSegments"] = (int64_t)ckp.maxSegments;

    ctx->saved_data["imgWidth"] = imgWidth; ctx->saved_data["imgHeight"] = imgHeight;
    ctx->saved_data["fx"] = fx; ctx->saved_data["fy"] = fy;
    ctx->saved_data["cx"] = cx; ctx->saved_data["cy"] = cy;
    ctx->saved_data["flags"] = (int64_t)flags;
    ctx->saved_data["K"] = K; ctx->saved_data["degreesToUse"] = degreesToUse;
    Tensor gradOut = (xysGradOut.has_value() && xysGradOut->defined()) ? *xysGradOut : Tensor();
    if (gradOut.defined()) {
        GS_CHECK_DEV(gradOut); GS_CHECK_F32(gradOut);
        TORCH_CHECK(gradOut.is_contiguous() && gradOut.numel() == 2 * N, "xysGradOut must be a contiguous [N,2] tensor");
    }
    ctx->save_for_backward({means, logScales, quats, vmHold, pmHold, radii, rgbRaw, idsSorted,
                            tileBins, packed, finalTs, finalIdx, imgRaw,
                            gradOut.defined() ? gradOut : torch::empty({0}, f32), bgHold, cpHold,
                            tileOrder, opacityLogits, b.blockMasks, ckp.saved(f32)});
    Tensor xysOut = xys.detach();
    ctx->mark_non_differentiable({xysOut, radii});
    return {img, xysOut, radii};
}

tensor_list SplatRender::backward(AutogradContext *ctx, tensor_list grad_outputs) {
    variable_list sv = ctx->get_saved_variables();
    Tensor means = sv[0], logScales = sv[1], quats = sv[2], viewMat = sv[3], projMat = sv[4];
    Tensor radii = sv[5], rgbRaw = sv[6], idsSorted = sv[7], tileBins = sv[8], packed = sv[9];
    Tensor finalTs = sv[10], finalIdx = sv[11], imgRaw = sv[12], gradOut = sv[13], bgHold = sv[14], cpHold = sv[15], tileOrder = sv[16];
    const int64_t N = means.size(0), K = ctx->saved_data["K"].toInt();
    const int W = (int)ctx->saved_data["imgWidth"].toInt(), H = (int)ctx->saved_data["imgHeight"].toInt();
    c10::DeviceGuard guard(means.device());
    gs_stream_t s = current_stream();
    Tensor v_img = grad_outputs[0].contiguous();
    GS_CHECK_F32(v_img);
    const float *bg = bgHold.data_ptr<float>();
    const float *cp = cpHold.data_ptr<float>();
    auto f32 = means.options();
    Tensor opacityLogits = sv[17], blockMasks = sv[18], checkpoints = sv[19];
    const int32_t listStats[2] = {(int32_t)ctx->saved_data["listM"].toInt(),
                                  (int32_t)ctx->saved_data["listLongest"].toInt()};
    Tensor v_xy = (gradOut.numel() == 2 * N && N > 0) ? gradOut.view({N, 2}) : Tensor();
    const size_t wsBytes = gs_rasterize_backward_workspace_bytes((int)N);
    Tensor ws = torch::empty({(int64_t)(wsBytes ? wsBytes : 64)}, f32.dtype(torch::kUInt8));
    const uint32_t flags = (uint32_t)ctx->saved_data["flags"].toInt();
    const size_t ckBytes = (size_t)checkpoints.numel();
    // compositing backward: gradients stay in the 64-byte records of `ws` ...
    check_status(gs_rasterize_backward_ckpt(W, H, (int)N, idsSorted.data_ptr<int32_t>(),
                                            maskptr(blockMasks),
                                            tileBins.data_ptr<int32_t>(), fptr(packed), bg, fptr(finalTs),
                                            finalIdx.data_ptr<int32_t>(), fptr(v_img), nullptr,
                                            fptr(imgRaw), nullptr, nullptr, nullptr, nullptr, ws.data_ptr(),
                                            wsBytes, listStats,
                                            tileOrder.numel() ? tileOrder.data_ptr<int32_t>() : nullptr,
                                            flags | GS_FLAG_KEEP_RECORDS,
                                            ckBytes ? checkpoints.data_ptr() : nullptr, ckBytes,
                                            (int32_t)ctx->saved_data["segLen"].toInt(),
                                            (int32_t)ctx->saved_data["maxSegments"].toInt(), s),
                 "gs_rasterize_backward");
    // ... and one fused stage turns them into the six parameter gradients (gs_gaussian_backward)
    GsCamera cam = make_camera(ctx->saved_data["fx"].toDouble(), ctx->saved_data["fy"].toDouble(),
                               ctx->saved_data["cx"].toDouble(), ctx->saved_data["cy"].toDouble(), H, W,
                               0.01, 1.0, GS_CAM_LOG_SCALES);
    Tensor vmHold, pmHold;
    const float *vmDev = matrix_arg(viewMat, vmHold, cam.viewmat);
    const float *pmDev = matrix_arg(projMat, pmHold, cam.projmat);
    Tensor v_means = torch::empty({N, 3}, f32), v_scales = torch::empty({N, 3}, f32);
    Tensor v_quats = torch::empty({N, 4}, f32), v_opacity = torch::empty({N, 1}, f32);
    Tensor v_dc = torch::empty({N, 3}, f32);
    Tensor v_rest = K > 1 ? torch::empty({N, K - 1, 3}, f32) : Tensor();
    if (N > 0)
        check_status(gs_gaussian_backward(&cam, vmDev, pmDev, (int)N, (int)K,
                                          (int)ctx->saved_data["degreesToUse"].toInt(), fptr(means),
                                          fptr(logScales), fptr(quats), fptr(opacityLogits), cp,
                                          radii.data_ptr<int32_t>(), fptr(rgbRaw), ws.data_ptr(), wsBytes,
                                          fptr_mut(v_means), fptr_mut(v_scales), fptr_mut(v_quats),
                                          fptr_mut(v_opacity), fptr_mut(v_dc),
                                          K > 1 ? fptr_mut(v_rest) : nullptr,
                                          v_xy.defined() ? fptr_mut(v_xy) : nullptr, flags, s),
                     "gs_gaussian_backward");
    Tensor none;
    return {v_means, v_scales, v_quats, v_opacity, v_dc, v_rest, none, none, none, none, none, none,
            none, none, none, none, none, none};
}

// ---- row f2: loss + optimiser (include/gsplat_train.h) -------------------------------------------
Tensor MainLoss::forward(AutogradContext *ctx, Tensor rgb, Tensor gt, double ssimWeight) {
    GS_CHECK_DEV(rgb); GS_CHECK_DEV(gt); GS_CHECK_F32(rgb); GS_CHECK_F32(gt);
    TORCH_CHECK(rgb.dim() == 3 && rgb.size(2) == 3, "rgb must be [H, W, 3]");
    TORCH_CHECK(gt.sizes() == rgb.sizes(), "gt must have rgb's shape");
    c10::DeviceGuard guard(rgb.device());
    Tensor r = rgb.contiguous(), g = gt.contiguous();
    const int H = (int)r.size(0), W = (int)r.size(1);
    Tensor loss = torch::empty({3}, r.options());
    Tensor vRgb = torch::empty_like(r);
    Tensor ws = torch::empty({(int64_t)gs_loss_workspace_bytes(W, H)}, r.options().dtype(torch::kUInt8));
    check_status(gs_main_loss(W, H, fptr(r), fptr(g), (float)ssimWeight, 1.0f, fptr_mut(loss),
                              fptr_mut(vRgb), ws.data_ptr(), (size_t)ws.numel(), current_stream()),
                 "gs_main_loss");
    ctx->save_for_backward({vRgb});
    return loss[0];
}

tensor_list MainLoss::backward(AutogradContext *ctx, tensor_list grad_outputs) {
    Tensor vRgb = ctx->get_saved_variables()[0];
    c10::DeviceGuard guard(vRgb.device());
    return {vRgb * grad_outputs[0], Tensor(), Tensor()};
}

Tensor mainLoss(const Tensor &rgb, const Tensor &gt, float ssimWeight) {
    return MainLoss::apply(rgb, gt, (double)ssimWeight);
}

FusedAdam::FusedAdam(std::vector<Tensor> params, std::vector<double> lrs)
    : params_(std::move(params)), lrs_(std::move(lrs)) {
    TORCH_CHECK(params_.size() == lrs_.size(), "one learning rate per parameter group");
    TORCH_CHECK(params_.size() <= GS_ADAM_MAX_GROUPS, "at most ", GS_ADAM_MAX_GROUPS, " groups");
    for (auto &p : params_) {
        GS_CHECK_DEV(p); GS_CHECK_F32(p);
        TORCH_CHECK(p.is_contiguous(), "parameters must be contiguous");
        expAvg_.push_back(torch::zeros_like(p));     // AdamParamState::exp_avg / exp_avg_sq
        expAvgSq_.push_back(torch::zeros_like(p));
    }
}

void FusedAdam::step() {
    if (params_.empty()) return;
    c10::DeviceGuard guard(params_[0].device());
    GsAdamGroup groups[GS_ADAM_MAX_GROUPS];
    std::vector<Tensor> keep;  // contiguous gradient copies, alive until the launch is enqueued
    int n = 0;
    for (size_t i = 0; i < params_.size(); i++) {
        const Tensor &g = params_[i].grad();
        if (!g.defined()) continue;  // torch::optim::Adam skips parameters without a gradient
        Tensor gc = g.contiguous();
        keep.push_back(gc);
        groups[n].param = params_[i].data_ptr<float>();
        groups[n].grad = gc.data_ptr<float>();
        groups[n].exp_avg = expAvg_[i].data_ptr<float>();
        groups[n].exp_avg_sq = expAvgSq_[i].data_ptr<float>();
        groups[n].n = params_[i].numel();
        groups[n].lr = lrs_[i];
        n++;
    }
    step_++;
    check_status(gs_adam_step(n, groups, step_, 0.9, 0.999, 1e-8, current_stream()), "gs_adam_step");
}

void FusedAdam::zeroGrad() {
    for (auto &p : params_)
        if (p.grad().defined()) p.mutable_grad().reset();  // set_to_none, Optimizer::zero_grad's default
}

void FusedAdam::replaceParam(size_t group, Tensor param, Tensor expAvg, Tensor expAvgSq) {
    GS_CHECK_DEV(param); GS_CHECK_F32(param);
    TORCH_CHECK(expAvg.sizes() == param.sizes() && expAvgSq.sizes() == param.sizes(),
                "optimiser state must have the parameter's shape");
    params_.at(group) = std::move(param);
    expAvg_.at(group) = expAvg.contiguous();
    expAvgSq_.at(group) = expAvgSq.contiguous();
}

float schedulerLearningRate(float lrInit, float lrFinal, int maxSteps, int step) {
    return gs_sched_lr(lrInit, lrFinal, maxSteps, step);
}

// ---- row f4: densification (include/gsplat_densify.h) ---------------------------------------------
void densifyStats(const Tensor &xysGrad, const Tensor &radii, int lastHeight, int lastWidth,
                  Tensor &xysGradNorm, Tensor &visCounts, Tensor &max2DSize) {
    GS_CHECK_DEV(xysGrad); GS_CHECK_F32(xysGrad); GS_CHECK_DEV(radii); GS_CHECK_I32(radii);
    const int64_t N = radii.numel();
    TORCH_CHECK(xysGrad.numel() == 2 * N, "xysGrad must be [N, 2]");
    c10::DeviceGuard guard(radii.device());
    const bool first = !xysGradNorm.numel();   // model.cpp:321,329
    if (first) {
        xysGradNorm = torch::empty({N}, xysGrad.options());
        visCounts = torch::empty({N}, xysGrad.options());
        max2DSize = torch::empty({N}, xysGrad.options());
    }
    Tensor g = xysGrad.contiguous(), r = radii.contiguous();
    check_status(gs_densify_stats((int)N, fptr(g), r.data_ptr<int32_t>(),
                                  (float)std::max(lastHeight, lastWidth), first ? 1 : 0,
                                  fptr_mut(xysGradNorm), fptr_mut(visCounts), fptr_mut(max2DSize),
                                  current_stream()),
                 "gs_densify_stats");
}

namespace {
GsGaussianSet as_set(const std::vector<Tensor> &t) {
    GsGaussianSet s = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (t.size() != 6) return s;
    float **f[6] = {&s.means, &s.log_scales, &s.quats, &s.opacity_logits, &s.features_dc, &s.features_rest};
    for (int i = 0; i < 6; i++)
        if (t[i].defined() && t[i].numel() > 0) *f[i] = const_cast<float *>(t[i].data_ptr<float>());
    return s;
}
}  // namespace

DensifyResult densify(const std::vector<Tensor> &params, const std::vector<Tensor> &expAvg,
                      const std::vector<Tensor> &expAvgSq, const Tensor &xysGradNorm,
                      const Tensor &visCounts, const Tensor &max2DSize, int lastWidth, int lastHeight,
                      float densifyGradThresh, float densifySizeThresh, bool checkScreenSize,
                      float splitScreenSize, bool cullHuge) {
    TORCH_CHECK(params.size() == 6, "six parameter tensors expected");
    TORCH_CHECK(expAvg.empty() || expAvg.size() == 6, "six exp_avg tensors (or none) expected");
    TORCH_CHECK(expAvgSq.size() == expAvg.size(), "exp_avg / exp_avg_sq must come together");
    for (auto &p : params) { GS_CHECK_DEV(p); GS_CHECK_F32(p); TORCH_CHECK(p.is_contiguous(), "contiguous parameters expected"); }
    const int N = (int)params[0].size(0);
    const int K = 1 + (int)(params[5].numel() > 0 ? params[5].size(1) : 0);
    c10::DeviceGuard guard(params[0].device());
    const GsDensifyConfig cfg = {0.5f * (float)std::max(lastWidth, lastHeight), densifyGradThresh,
                                 densifySizeThresh, splitScreenSize, checkScreenSize ? 1 : 0, 0.1f,
                                 cullHuge ? 1 : 0, 0.5f, 0.15f};
    auto bytes = gs_densify_workspace_bytes(N);
    Tensor ws = torch::empty({(int64_t)bytes}, params[0].options().dtype(torch::kUInt8));
    Tensor counts = torch::zeros({8}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
    check_status(gs_densify_plan(N, &cfg, fptr(xysGradNorm), fptr(visCounts), fptr(max2DSize),
                                 fptr(params[1]), fptr(params[3]), counts.data_ptr<int32_t>(),
                                 ws.data_ptr(), bytes, current_stream()),
                 "gs_densify_plan");
    c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().synchronize();   // the one sync: nSplits, newN
    const int32_t *c = counts.data_ptr<int32_t>();
    DensifyResult r;
    r.nSplits = c[GS_DENSIFY_N_SPLITS]; r.nDups = c[GS_DENSIFY_N_DUPS];
    r.added = c[GS_DENSIFY_ADDED]; r.culled = c[GS_DENSIFY_CULLED];
    const int64_t newN = c[GS_DENSIFY_NEW_N];
    Tensor samples;
    if (r.nSplits > 0) samples = torch::randn({2 * (int64_t)r.nSplits, 3}, params[0].options());  // model.cpp:360
    auto alloc = [&](std::vector<Tensor> &dst) {
        for (auto &p : params) {
            auto sizes = p.sizes().vec();
            sizes[0] = newN;
            dst.push_back(torch::empty(sizes, p.options()));
        }
    };
    alloc(r.params);
    if (!expAvg.empty()) { alloc(r.expAvg); alloc(r.expAvgSq); }
    const GsGaussianSet src[3] = {as_set(params), as_set(expAvg), as_set(expAvgSq)};
    const GsGaussianSet dst[3] = {as_set(r.params), as_set(r.expAvg), as_set(r.expAvgSq)};
    check_status(gs_densify_apply(N, K, (int)newN, samples.defined() ? fptr(samples) : nullptr, src, dst,
                                  ws.data_ptr(), bytes, current_stream()),
                 "gs_densify_apply");
    return r;
}

void resetOpacity(Tensor &opacities, float resetValue, c10::optional<Tensor> expAvg,
                  c10::optional<Tensor> expAvgSq) {
    GS_CHECK_DEV(opacities); GS_CHECK_F32(opacities);
    TORCH_CHECK(opacities.is_contiguous(), "opacities must be contiguous");
    c10::DeviceGuard guard(opacities.device());
    check_status(gs_reset_opacity((int)opacities.numel(), resetValue, fptr_mut(opacities),
                                  expAvg ? expAvg->data_ptr<float>() : nullptr,
                                  expAvgSq ? expAvgSq->data_ptr<float>() : nullptr, current_stream()),
                 "gs_reset_opacity");
}

// ---- Python-visible registration (torch.ops.opensplat_amd.*) -------------------------------------
namespace {

std::vector<Tensor> op_project_gaussians(const Tensor &means, const Tensor &scales, double globScale,
                                         const Tensor &quats, const Tensor &viewMat,
                                         const Tensor &projMat, double fx, double fy, double cx,
                                         double cy, int64_t imgHeight, int64_t imgWidth,
                                         double clipThresh) {
    TileBounds tb = std::make_tuple((int)((imgWidth + BLOCK_X - 1) / BLOCK_X),
                                    (int)((imgHeight + BLOCK_Y - 1) / BLOCK_Y), 1);
    return ProjectGaussians::apply(means, scales, globScale, quats, viewMat, projMat, fx, fy, cx, cy,
                                   imgHeight, imgWidth, tb, clipThresh);
}

Tensor op_rasterize_gaussians(const Tensor &xys, const Tensor &depths, const Tensor &radii,
                              const Tensor &conics, const Tensor &numTilesHit, const Tensor &colors,
                              const Tensor &opacity, int64_t imgHeight, int64_t imgWidth,
                              const Tensor &background, const c10::optional<Tensor> &cov2d) {
    return RasterizeGaussians::apply(xys, depths, radii, conics, numTilesHit, colors, opacity,
                                     imgHeight, imgWidth, background, cov2d);
}

Tensor op_spherical_harmonics(int64_t degreesToUse, const Tensor &viewDirs, const Tensor &coeffs) {
    return SphericalHarmonics::apply(degreesToUse, viewDirs, coeffs);
}

std::vector<Tensor> op_splat_render(const Tensor &means, const Tensor &logScales, const Tensor &quats,
                                    const Tensor &opacityLogits, const Tensor &featuresDc,
                                    const Tensor &featuresRest, const Tensor &viewMat,
                                    const Tensor &projMat, const Tensor &camPos, double fx, double fy,
                                    double cx, double cy, int64_t imgHeight, int64_t imgWidth,
                                    int64_t degreesToUse, const Tensor &background,
                                    const c10::optional<Tensor> &xysGradOut) {
    return SplatRender::apply(means, logScales, quats, opacityLogits, featuresDc, featuresRest, viewMat,
                              projMat, camPos, fx, fy, cx, cy, imgHeight, imgWidth, degreesToUse,
                              background, xysGradOut);
}

void op_set_fast_exp(bool enabled) { gsplatSetFastExp(enabled); }
void op_set_segmented_backward(bool enabled) { gsplatSetSegmentedBackward(enabled); }

// -> [params x6, exp_avg x6, exp_avg_sq x6 (moment lists empty when none were given), counts int32[4]
//     = {nSplits, nDups, added, culled} on the host]
std::vector<Tensor> op_densify(std::vector<Tensor> params, std::vector<Tensor> expAvg,
                               std::vector<Tensor> expAvgSq, const Tensor &xysGradNorm,
                               const Tensor &visCounts, const Tensor &max2DSize, int64_t lastWidth,
                               int64_t lastHeight, double densifyGradThresh, double densifySizeThresh,
                               bool checkScreenSize, double splitScreenSize, bool cullHuge) {
    DensifyResult r = densify(params, expAvg, expAvgSq, xysGradNorm, visCounts, max2DSize,
                              (int)lastWidth, (int)lastHeight, (float)densifyGradThresh,
                              (float)densifySizeThresh, checkScreenSize, (float)splitScreenSize, cullHuge);
    std::vector<Tensor> out = r.params;
    out.insert(out.end(), r.expAvg.begin(), r.expAvg.end());
    out.insert(out.end(), r.expAvgSq.begin(), r.expAvgSq.end());
    Tensor counts = torch::empty({4}, torch::kInt32);
    int32_t *c = counts.data_ptr<int32_t>();
    c[0] = r.nSplits; c[1] = r.nDups; c[2] = r.added; c[3] = r.culled;
    out.push_back(counts);
    return out;
}

std::vector<Tensor> op_densify_stats(const Tensor &xysGrad, const Tensor &radii, int64_t lastHeight,
                                     int64_t lastWidth, Tensor xysGradNorm, Tensor visCounts,
                                     Tensor max2DSize) {
    densifyStats(xysGrad, radii, (int)lastHeight, (int)lastWidth, xysGradNorm, visCounts, max2DSize);
    return {xysGradNorm, visCounts, max2DSize};
}

Tensor op_main_loss(const Tensor &rgb, const Tensor &gt, double ssimWeight) {
    return MainLoss::apply(rgb, gt, ssimWeight);
}

// One optimizersStep over parallel lists (params updated in place; state tensors owned by the
// caller so that Python can hold them): the FusedAdam class is the C++ face of the same call.
void op_adam_step(std::vector<Tensor> params, std::vector<Tensor> grads, std::vector<Tensor> expAvg,
                  std::vector<Tensor> expAvgSq, std::vector<double> lrs, int64_t step) {
    fusedAdamStep(params, grads, expAvg, expAvgSq, lrs, step);
}
}  // namespace

void fusedAdamStep(const std::vector<Tensor> &params, const std::vector<Tensor> &grads,
                   const std::vector<Tensor> &expAvg, const std::vector<Tensor> &expAvgSq,
                   const std::vector<double> &lrs, int64_t step) {
    const size_t n = params.size();
    TORCH_CHECK(grads.size() == n && expAvg.size() == n && expAvgSq.size() == n && lrs.size() == n,
                "parallel lists of equal length expected");
    TORCH_CHECK(n <= GS_ADAM_MAX_GROUPS, "at most ", GS_ADAM_MAX_GROUPS, " groups");
    if (n == 0) return;
    c10::DeviceGuard guard(params[0].device());
    GsAdamGroup groups[GS_ADAM_MAX_GROUPS];
    for (size_t i = 0; i < n; i++) {
        for (const Tensor *t : {&params[i], &grads[i], &expAvg[i], &expAvgSq[i]}) {
            TORCH_CHECK(t->is_cuda() && t->scalar_type() == torch::kFloat32 && t->is_contiguous(),
                        "contiguous float32 GPU tensors expected");
            TORCH_CHECK(t->numel() == params[i].numel(), "group ", i, ": size mismatch");
        }
        groups[i] = GsAdamGroup{params[i].data_ptr<float>(), grads[i].data_ptr<float>(),
                                expAvg[i].data_ptr<float>(), expAvgSq[i].data_ptr<float>(),
                                params[i].numel(), lrs[i]};
    }
    check_status(gs_adam_step((int)n, groups, step, 0.9, 0.999, 1e-8, current_stream()), "gs_adam_step");
}

namespace {

}  // namespace

// ---- GradExchange (include/gsplat_dist.h) --------------------------------------------------------
static void check_dist(int rc, const char *what) {
    TORCH_CHECK(rc == GS_OK, what, " failed: ", gs_strerror(rc), " — ", gs_dist_last_error());
}
std::vector<uint8_t> GradExchange::uniqueId() {
    std::vector<uint8_t> id(GS_DIST_ID_BYTES);
    check_dist(gs_dist_unique_id(id.data()), "gs_dist_unique_id");
    return id;
}
GradExchange::GradExchange(int worldSize, int rank, const std::vector<uint8_t> &id, int device) {
    TORCH_CHECK(id.size() == GS_DIST_ID_BYTES, "GradExchange: id must hold ", GS_DIST_ID_BYTES, " bytes");
    GsDistComm *c = nullptr;
    check_dist(gs_dist_init(&c, worldSize, rank, id.data(), device), "gs_dist_init");
    comm_ = c;
}
GradExchange::~GradExchange() { (void)gs_dist_destroy(static_cast<GsDistComm *>(comm_)); }
int GradExchange::worldSize() const { return gs_dist_world_size(static_cast<const GsDistComm *>(comm_)); }
int GradExchange::rank() const { return gs_dist_rank(static_cast<const GsDistComm *>(comm_)); }
void GradExchange::allReduce(Tensor flat) {
    GS_CHECK_DEV(flat);
    GS_CHECK_F32(flat);
    TORCH_CHECK(flat.is_contiguous(), "GradExchange: contiguous buffer required");
    c10::DeviceGuard guard(flat.device());
    check_dist(gs_dist_allreduce_sum(static_cast<GsDistComm *>(comm_), flat.data_ptr<float>(),
                                     (size_t)flat.numel(), current_stream()), "gs_dist_allreduce_sum");
}
void GradExchange::allReduceBuckets(Tensor flat, int nBuckets) {
    GS_CHECK_DEV(flat);
    GS_CHECK_F32(flat);
    TORCH_CHECK(flat.is_contiguous(), "GradExchange: contiguous buffer required");
    c10::DeviceGuard guard(flat.device());
    check_dist(gs_dist_allreduce_sum_buckets(static_cast<GsDistComm *>(comm_), flat.data_ptr<float>(),
                                             (size_t)flat.numel(), nBuckets, nullptr, current_stream()),
               "gs_dist_allreduce_sum_buckets");
}
void GradExchange::allGather(Tensor message, Tensor gathered) {
    GS_CHECK_DEV(message);
    GS_CHECK_DEV(gathered);
    GS_CHECK_F32(message);
    GS_CHECK_F32(gathered);
    TORCH_CHECK(message.is_contiguous() && gathered.is_contiguous(), "GradExchange: contiguous buffers required");
    TORCH_CHECK(gathered.numel() == message.numel() * (int64_t)worldSize(),
                "GradExchange::allGather: gathered must hold worldSize messages");
    c10::DeviceGuard guard(message.device());
    check_dist(gs_dist_allgather(static_cast<GsDistComm *>(comm_), message.data_ptr<float>(),
                                 gathered.data_ptr<float>(), (size_t)message.numel(), current_stream()),
               "gs_dist_allgather");
}
void GradExchange::exchangeFactored(Tensor geometry, Tensor message, Tensor gathered, Tensor means,
                                    int degreesToUse, Tensor v_dc, Tensor v_rest) {
    GS_CHECK_DEV(means);
    GS_CHECK_F32(means);
    GS_CHECK_DEV(v_dc);
    GS_CHECK_F32(v_dc);
    TORCH_CHECK(means.is_contiguous() && v_dc.is_contiguous(), "GradExchange: contiguous tensors required");
    const int64_t N = means.size(0);
    TORCH_CHECK(message.numel() == 4 + 3 * N, "GradExchange: message is [camera centre 4 | v_colour N x 3]");
    const int K = v_rest.defined() && v_rest.numel() > 0 ? (int)v_rest.size(1) + 1 : 1;
    if (K > 1) {
        GS_CHECK_DEV(v_rest);
        GS_CHECK_F32(v_rest);
        TORCH_CHECK(v_rest.is_contiguous() && v_rest.size(0) == N && v_rest.size(2) == 3, "v_rest must be [N, K-1, 3]");
    }
    allReduce(geometry);
    allGather(message, gathered);
    c10::DeviceGuard guard(means.device());
    const int rc = gs_sh_backward_cameras((int)N, K, degreesToUse, worldSize(), means.data_ptr<float>(),
                                          gathered.data_ptr<float>(), (size_t)message.numel(),
                                          gathered.data_ptr<float>() + 4, (size_t)message.numel(),
                                          v_dc.data_ptr<float>(), K > 1 ? v_rest.data_ptr<float>() : nullptr,
                                          0u, current_stream());
    TORCH_CHECK(rc == GS_OK, "gs_sh_backward_cameras failed: ", gs_strerror(rc), " — ", gs_last_hip_error());
}
// self-test ops: a one-rank communicator through the whole C++ / C-ABI / RCCL stack
static Tensor op_grad_exchange_selftest(Tensor flat, int64_t buckets) {
    GradExchange ex(1, 0, GradExchange::uniqueId(), flat.device().index());
    if (buckets > 1) ex.allReduceBuckets(flat, (int)buckets); else ex.allReduce(flat);
    TORCH_CHECK(ex.worldSize() == 1 && ex.rank() == 0);
    return flat;
}
static std::vector<Tensor> op_grad_exchange_factored_selftest(Tensor geometry, Tensor message, Tensor means,
                                                              int64_t K, int64_t degrees_to_use) {
    GradExchange ex(1, 0, GradExchange::uniqueId(), means.device().index());
    const int64_t N = means.size(0);
    Tensor gathered = torch::empty_like(message);
    Tensor v_dc = torch::empty({N, 3}, means.options());
    Tensor v_rest = torch::empty({N, K - 1, 3}, means.options());
    ex.exchangeFactored(geometry, message, gathered, means, (int)degrees_to_use, v_dc, v_rest);
    return {v_dc, v_rest, gathered};
}

static void op_binning_reset() { gsplatResetBinningState(); }
static std::vector<int64_t> op_binning_counters() {
    auto c = gsplatBinningCounters();
    return {std::get<0>(c), std::get<1>(c)};
}
static std::vector<int64_t> op_cov2d_channel_counters(bool reset) {
    auto c = gsplatCov2dChannelCounters(reset);
    return {std::get<0>(c), std::get<1>(c)};
}
static int64_t op_binning_capacity(int64_t device, int64_t w, int64_t h) {
    return gsplatBinningCapacity((int)device, (int)w, (int)h);
}

static std::vector<Tensor> op_bin_and_sort_gaussians(int64_t numPoints, int64_t numIntersects, Tensor xys,
                                                     Tensor depths, Tensor radii, Tensor cumTilesHit,
                                                     int64_t tilesX, int64_t tilesY) {
    auto t = binAndSortGaussians((int)numPoints, (int)numIntersects, xys, depths, radii, cumTilesHit,
                                 std::make_tuple((int)tilesX, (int)tilesY, 1));
    return {std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t), std::get<4>(t)};
}


// ---- CameraBatch: a camera batch with two cameras in flight (gsplat_ops.hpp) -----------------------------
struct CameraBatch::Lane {
    c10::hip::HIPStreamMasqueradingAsCUDA stream;
    at::cuda::CUDAEvent done;
    // what one camera in flight owns
    Tensor packed, depths, radii, rgbRaw, imgRaw, img, finalTs, finalIdx, records;
    BinnedLists lists;
    GsCamera cam;
    Tensor vmHold, pmHold, cpHold;
    const float *vmDev = nullptr, *pmDev = nullptr, *cp = nullptr;
    int64_t N = -1, K = 0;
    bool det = false;
    explicit Lane(int device) : stream(c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, (c10::DeviceIndex)device)) {}
};

CameraBatch::CameraBatch(int64_t imgHeight, int64_t imgWidth) : H_(imgHeight), W_(imgWidth) {}
CameraBatch::~CameraBatch() = default;

void CameraBatch::forwardBackward(const Tensor &meansIn, const Tensor &logScalesIn, const Tensor &quatsIn,
                                  const Tensor &opacityLogitsIn, const Tensor &featuresDcIn,
                                  const Tensor &featuresRestIn, const std::vector<BatchCamera> &cameras,
                                  int64_t degreesToUse, const Tensor &background, const Cotangent &cotangent,
                                  std::vector<Tensor> &grads, bool deterministic, bool serial,
                                  GradExchange *exchange) {
    TORCH_CHECK(!cameras.empty(), "CameraBatch: no camera");
    GS_CHECK_DEV(meansIn); GS_CHECK_F32(meansIn);
    const int64_t N = meansIn.size(0);
    const bool hasRest = featuresRestIn.defined() && featuresRestIn.numel() > 0;
    const int64_t K = 1 + (hasRest ? featuresRestIn.size(1) : 0);
    TORCH_CHECK(grads.size() >= 5 && (K == 1 || grads.size() >= 6), "CameraBatch: six gradient tensors expected");
    for (size_t i = 0; i < grads.size(); i++)
        if (i < 5 || K > 1) {
            GS_CHECK_DEV(grads[i]); GS_CHECK_F32(grads[i]);
            TORCH_CHECK(grads[i].is_contiguous(), "CameraBatch: gradient tensors must be contiguous");
        }
    TORCH_CHECK(grads[0].numel() == 3 * N && grads[1].numel() == 3 * N && grads[2].numel() == 4 * N &&
                grads[3].numel() == N && grads[4].numel() == 3 * N && (K == 1 || grads[5].numel() == 3 * (K - 1) * N),
                "CameraBatch: gradient tensors do not match the parameters");
    c10::DeviceGuard guard(meansIn.device());
    const int device = meansIn.get_device();
    Tensor means = meansIn.contiguous(), logScales = logScalesIn.contiguous(), quats = quatsIn.contiguous();
    Tensor opacityLogits = opacityLogitsIn.contiguous(), featuresDc = featuresDcIn.contiguous();
    Tensor featuresRest = hasRest ? featuresRestIn.contiguous() : Tensor();
    const int W = (int)W_, H = (int)H_;
    auto f32 = means.options();
    auto i32 = means.options().dtype(torch::kInt32);
    Tensor bgHold;
    const float *bg = vec3_arg(background, bgHold);
    const uint32_t flags = (g_fast_exp.load() ? GS_FLAG_FAST_EXP : 0u) | GS_FLAG_CLAMP_IMAGE | GS_FLAG_LOGIT_OPACITY;
    const uint32_t keep = GS_FLAG_KEEP_RECORDS | (deterministic ? GS_FLAG_DETERMINISTIC : 0u);
    const size_t wsBytes = deterministic ? gs_rasterize_backward_workspace_bytes_det((int)N)
                                         : gs_rasterize_backward_workspace_bytes((int)N);
    auto mainStream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA();
    at::cuda::CUDAEvent start;
    start.record(mainStream);
    for (auto &lp : lanes_) {
        if (!lp) lp = std::make_unique<Lane>(device);
        Lane &L = *lp;
        c10::hip::HIPStreamGuardMasqueradingAsCUDA sg(L.stream);   // (allocations belong to the lane's stream)
        if (L.N != N || L.K != K || L.det != deterministic) {
            L.packed = torch::empty({N, GS_SPLAT_DWORDS}, f32); L.depths = torch::empty({N}, f32);
            L.radii = torch::empty({N}, i32); L.rgbRaw = torch::empty({N, 3}, f32);
            L.imgRaw = torch::empty({H, W, 3}, f32); L.img = torch::empty({H, W, 3}, f32);
            L.finalTs = torch::empty({H, W}, f32); L.finalIdx = torch::empty({H, W}, i32);
            L.records = torch::zeros({(int64_t)(wsBytes ? wsBytes : 64)}, f32.dtype(torch::kUInt8));
            L.N = N; L.K = K; L.det = deterministic;
        }
        start.block(L.stream);
    }
    auto front = [&](Lane &L, int j) {
        const BatchCamera &c = cameras[(size_t)j];
        c10::hip::HIPStreamGuardMasqueradingAsCUDA sg(L.stream);
        gs_stream_t s = current_stream();
        L.cam = make_camera(c.fx, c.fy, c.cx, c.cy, H_, W_, 0.01, 1.0, GS_CAM_LOG_SCALES);
        L.vmDev = matrix_arg(c.viewMat, L.vmHold, L.cam.viewmat);
        L.pmDev = matrix_arg(c.projMat, L.pmHold, L.cam.projmat);
        L.cp = vec3_arg(c.camPos, L.cpHold);
        check_status(gs_gaussian_forward(&L.cam, L.vmDev, L.pmDev, (int)N, (int)K, (int)degreesToUse, fptr(means),
                                         fptr(logScales), fptr(quats), fptr(opacityLogits), fptr(featuresDc),
                                         hasRest ? fptr(featuresRest) : nullptr, L.cp, fptr_mut(L.packed),
                                         fptr_mut(L.depths), L.radii.data_ptr<int32_t>(), fptr_mut(L.rgbRaw),
                                         nullptr, flags, s),
                     "gs_gaussian_forward");
        L.lists = binPackedRecords(L.packed, L.depths, H, W);
        check_status(gs_rasterize_forward(W, H, L.lists.gaussianIdsSorted.data_ptr<int32_t>(), maskptr(L.lists.blockMasks),
                                          L.lists.tileBins.data_ptr<int32_t>(), fptr(L.packed), bg, fptr_mut(L.imgRaw),
                                          fptr_mut(L.finalTs), L.finalIdx.data_ptr<int32_t>(), fptr_mut(L.img), nullptr,
                                          L.lists.tileOrder.data_ptr<int32_t>(), flags, s),
                     "gs_rasterize_forward");
    };
    auto back = [&](Lane &L, int j, Lane *prev) {
        c10::hip::HIPStreamGuardMasqueradingAsCUDA sg(L.stream);
        gs_stream_t s = current_stream();
        Tensor v = cotangent(j, L.img).contiguous();
        GS_CHECK_DEV(v); GS_CHECK_F32(v);
        TORCH_CHECK(v.numel() == (int64_t)H * W * 3, "CameraBatch: the cotangent must be [H,W,3]");
        check_status(gs_rasterize_backward(W, H, (int)N, L.lists.gaussianIdsSorted.data_ptr<int32_t>(),
                                           maskptr(L.lists.blockMasks), L.lists.tileBins.data_ptr<int32_t>(),
                                           fptr(L.packed), bg, fptr(L.finalTs), L.finalIdx.data_ptr<int32_t>(), fptr(v),
                                           nullptr, fptr(L.imgRaw), nullptr, nullptr, nullptr, nullptr,
                                           L.records.data_ptr(), wsBytes, L.lists.listStats,
                                           L.lists.tileOrder.data_ptr<int32_t>(), flags | keep, s),
                     "gs_rasterize_backward");
        if (prev) prev->done.block(L.stream);     // the gradient tensors: camera order
        check_status(gs_gaussian_backward(&L.cam, L.vmDev, L.pmDev, (int)N, (int)K, (int)degreesToUse, fptr(means),
                                          fptr(logScales), fptr(quats), fptr(opacityLogits), L.cp,
                                          L.radii.data_ptr<int32_t>(), fptr(L.rgbRaw), L.records.data_ptr(), wsBytes,
                                          fptr_mut(grads[0]), fptr_mut(grads[1]), fptr_mut(grads[2]),
                                          fptr_mut(grads[3]), fptr_mut(grads[4]), K > 1 ? fptr_mut(grads[5]) : nullptr,
                                          nullptr, flags | (j > 0 ? GS_FLAG_ACCUMULATE_GRADS : 0u), s),
                     "gs_gaussian_backward");
        L.done.record(L.stream);
    };
    const int n = (int)cameras.size();
    Lane *prev = nullptr;
    if (N > 0) {
        if (!serial) front(*lanes_[0], 0);
        for (int j = 0; j < n; j++) {
            Lane &L = *lanes_[serial ? 0 : j % 2];
            if (serial) front(L, j);
            while (!validateBinning(L.lists)) front(L, j);     // (the id list was too small: this camera's front again)
            lastM_ = L.lists.listStats[0];
            if (!serial && j + 1 < n) front(*lanes_[(j + 1) % 2], j + 1);
            back(L, j, prev);
            prev = &L;
        }
        prev->done.block(mainStream);
    }
    if (exchange && exchange->worldSize() > 1)
        for (size_t i = 0; i < grads.size(); i++)
            if (i < 5 || K > 1) exchange->allReduce(grads[i]);
}

// test face: fixed cotangents v_out [c,H,W,3]; -> { v_means, v_logScales, v_quats, v_opacityLogits, v_dc, v_rest, rgb [c,H,W,3] }
static std::vector<Tensor> op_camera_batch_step(const Tensor &means, const Tensor &logScales, const Tensor &quats,
                                                const Tensor &opacityLogits, const Tensor &featuresDc,
                                                const Tensor &featuresRest, const Tensor &viewMats,
                                                const Tensor &projMats, const Tensor &camPos, double fx, double fy,
                                                double cx, double cy, int64_t imgHeight, int64_t imgWidth,
                                                int64_t degreesToUse, const Tensor &background, const Tensor &vOut,
                                                bool deterministic, bool serial) {
    const int64_t c = viewMats.size(0), N = means.size(0);
    TORCH_CHECK(projMats.size(0) == c && camPos.size(0) == c && vOut.size(0) == c, "camera_batch_step: c cameras expected");
    std::vector<BatchCamera> cams((size_t)c);
    for (int64_t j = 0; j < c; j++) {
        cams[(size_t)j].viewMat = viewMats[j]; cams[(size_t)j].projMat = projMats[j]; cams[(size_t)j].camPos = camPos[j];
        cams[(size_t)j].fx = fx; cams[(size_t)j].fy = fy; cams[(size_t)j].cx = cx; cams[(size_t)j].cy = cy;
    }
    auto f32 = means.options();
    const bool hasRest = featuresRest.defined() && featuresRest.numel() > 0;
    std::vector<Tensor> grads = {torch::empty({N, 3}, f32), torch::empty({N, 3}, f32), torch::empty({N, 4}, f32),
                                 torch::empty({N}, f32), torch::empty({N, 3}, f32),
                                 hasRest ? torch::empty_like(featuresRest) : torch::empty({0}, f32)};
    Tensor rgb = torch::empty({c, imgHeight, imgWidth, 3}, f32);
    static std::mutex batchMutex;
    static std::map<std::tuple<int, int64_t, int64_t>, std::unique_ptr<CameraBatch>> batches;   // lanes re-used across calls
    std::lock_guard<std::mutex> lock(batchMutex);
    auto &cb = batches[std::make_tuple(means.get_device(), imgHeight, imgWidth)];
    if (!cb) cb = std::make_unique<CameraBatch>(imgHeight, imgWidth);
    cb->forwardBackward(means, logScales, quats, opacityLogits, featuresDc, featuresRest, cams, degreesToUse,
                        background, [&](int j, const Tensor &img) { rgb[j].copy_(img); return vOut[j]; }, grads,
                        deterministic, serial);
    grads.push_back(rgb);
    return grads;
}

// libgsplat_hip.so must be the one this file's header describes (shifted arguments otherwise).  Checked
// lazily by current_stream() in front of the first C-ABI call of the process (every consumer), by ops.py right
// after load_library (a readable Python error at import) and on request through gsplatCheckAbi(); NOT inside the
// static initialiser below: an exception thrown while dlopen runs ends in std::terminate.
static std::vector<int64_t> op_abi_versions() { return {(int64_t)gs_version(), (int64_t)GS_ABI_VERSION}; }
// (bindings_hip_native.cpp, the other translation unit of this library)
gs_stream_t gsplatCurrentStream() { return current_stream(); }
void gsplatCheckAbi() {
    TORCH_CHECK(gs_version() == GS_ABI_VERSION, "libgsplat_hip.so has ABI version ", gs_version(),
                ", libgsplat_torch.so was built against ", GS_ABI_VERSION, ": rebuild both");
}

TORCH_LIBRARY(opensplat_amd, m) {
    m.def("abi_versions() -> int[]", &op_abi_versions);
    m.def("bin_and_sort_gaussians(int num_points, int num_intersects, Tensor xys, Tensor depths, Tensor radii, "
          "Tensor cum_tiles_hit, int tiles_x, int tiles_y) -> Tensor[]", &op_bin_and_sort_gaussians);
    m.def("binning_reset() -> ()", &op_binning_reset);
    m.def("binning_counters() -> int[]", &op_binning_counters);
    m.def("binning_capacity(int device, int img_width, int img_height) -> int", &op_binning_capacity);
    m.def("cov2d_channel_counters(bool reset=False) -> int[]", &op_cov2d_channel_counters);
    m.def("grad_exchange_selftest(Tensor(a!) flat, int buckets) -> Tensor(a!)", &op_grad_exchange_selftest);
    m.def("grad_exchange_factored_selftest(Tensor(a!) geometry, Tensor message, Tensor means, int K, "
          "int degrees_to_use) -> Tensor[]", &op_grad_exchange_factored_selftest);
    m.def("project_gaussians(Tensor means, Tensor scales, float glob_scale, Tensor quats, "
          "Tensor viewmat, Tensor projmat, float fx, float fy, float cx, float cy, int img_height, "
          "int img_width, float clip_thresh=0.01) -> Tensor[]",
          &op_project_gaussians);
    m.def("rasterize_gaussians(Tensor xys, Tensor depths, Tensor radii, Tensor conics, "
          "Tensor num_tiles_hit, Tensor colors, Tensor opacity, int img_height, int img_width, "
          "Tensor background, Tensor? cov2d=None) -> Tensor",
          &op_rasterize_gaussians);
    m.def("spherical_harmonics(int degrees_to_use, Tensor viewdirs, Tensor coeffs) -> Tensor",
          &op_spherical_harmonics);
    m.def("splat_render(Tensor means, Tensor log_scales, Tensor quats, Tensor opacity_logits, "
          "Tensor features_dc, Tensor features_rest, Tensor viewmat, Tensor projmat, Tensor cam_pos, "
          "float fx, float fy, float cx, float cy, int img_height, int img_width, int degrees_to_use, "
          "Tensor background, Tensor? xys_grad_out=None) -> Tensor[]",
          &op_splat_render);
    m.def("camera_batch_step(Tensor means, Tensor log_scales, Tensor quats, Tensor opacity_logits, "
          "Tensor features_dc, Tensor features_rest, Tensor viewmats, Tensor projmats, Tensor cam_pos, float fx, "
          "float fy, float cx, float cy, int img_height, int img_width, int degrees_to_use, Tensor background, "
          "Tensor v_out, bool deterministic=False, bool serial=False) -> Tensor[]",
          &op_camera_batch_step);
    m.def("set_fast_exp(bool enabled) -> ()", &op_set_fast_exp);
    m.def("set_segmented_backward(bool enabled) -> ()", &op_set_segmented_backward);
    m.def("main_loss(Tensor rgb, Tensor gt, float ssim_weight) -> Tensor", &op_main_loss);
    m.def("densify_stats(Tensor xys_grad, Tensor radii, int last_height, int last_width, "
          "Tensor xys_grad_norm, Tensor vis_counts, Tensor max_2d_size) -> Tensor[]",
          &op_densify_stats);
    m.def("densify(Tensor[] params, Tensor[] exp_avg, Tensor[] exp_avg_sq, Tensor xys_grad_norm, "
          "Tensor vis_counts, Tensor max_2d_size, int last_width, int last_height, "
          "float densify_grad_thresh, float densify_size_thresh, bool check_screen_size, "
          "float split_screen_size, bool cull_huge) -> Tensor[]",
          &op_densify);
    m.def("adam_step(Tensor(a!)[] params, Tensor[] grads, Tensor(b!)[] exp_avg, Tensor(c!)[] exp_avg_sq, "
          "float[] lrs, int step) -> ()",
          &op_adam_step);
}
