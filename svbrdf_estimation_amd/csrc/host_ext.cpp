// host_ext.cpp -- native host path of one RenderingLoss / MixedLoss call.
//
// The Python host path (environment.BatchSceneSampler + ctypes + torch.autograd.Function) costs
// 160-220 us per call on the GPU box, 3x the fused kernel it launches.  This extension does the
// same three things -- draw the scenes (reference RNG order, development/multiImage_pytorch/
// losses.py:35 -> environment.py:18-55 -> utils.py:100-111), hand the [B,S,9] table to the kernel
// (by value with the launch when it has <= 288 rows, through a pinned upload ring otherwise), launch
// svbrdf_{mixed,head}_loss_fwd_bwd[_host_scenes] and hang the precomputed gradient on a C++
// autograd node -- without the Python interpreter in the loop.  It contains no arithmetic of
// the hot path: the kernels live in libsvbrdf_hip.so and are reached through the C ABI
// (include/svbrdf_hip.h), whose entry points are resolved with dlsym.  No HIP/ROCm headers are
// needed: the raw stream handle comes from Python, events use three runtime symbols by name.
//
// Built by __graft_entry__.build() with torch.utils.cpp_extension (plain C++ extension).

#include <torch/extension.h>
#include <torch/csrc/autograd/autograd.h>
#include <torch/csrc/autograd/graph_task.h>
#include <torch/csrc/autograd/anomaly_mode.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <ATen/CPUGeneratorImpl.h>
#include <ATen/core/DistributionsHelper.h>

#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

using loss_fn_t = int (*)(const float *, const float *, const float *, const float *, float, float, float, float *,
                          float *, void *, size_t, int, int, int, int, void *);
using render_fwd_fn_t = int (*)(const float *, const float *, int, const float *, float *, int, int, int, int, void *);
using render_bwd_fn_t = int (*)(const float *, const float *, int, const float *, const float *, float *, int, int, int,
                                int, void *);
using scale_fn_t = int (*)(float *, const float *, size_t, void *);
using ws_fn_t = size_t (*)(int, int, int, int);
using err_fn_t = const char *(*)();
using xrow_fn_t = int (*)(float *, int);
using ev_create_t = int (*)(void **, unsigned);
using ev_record_t = int (*)(void *, void *);
using ev_sync_t = int (*)(void *);

struct Abi {
    loss_fn_t mixed = nullptr;
    loss_fn_t head = nullptr;     // same signature, input = [B,9,H,W] encoded head output
    loss_fn_t mixed_host = nullptr, head_host = nullptr;   // scene table in host memory (kernel-argument block)
    int host_rows = 0;            // largest B*S those two take
    render_fwd_fn_t render_fwd_host = nullptr;             // K1 / K2 with the scene rows by value (svbrdf_hip.h)
    render_bwd_fn_t render_bwd_host = nullptr;
    scale_fn_t scale = nullptr;
    ws_fn_t ws_bytes = nullptr;
    err_fn_t last_error = nullptr;
    xrow_fn_t make_xrow = nullptr;
    ev_create_t ev_create = nullptr;
    ev_record_t ev_record = nullptr;
    ev_sync_t ev_sync = nullptr;
} g_abi;

void bind(const std::string &path)
{
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    TORCH_CHECK(h != nullptr, "cannot load ", path, ": ", dlerror());
    auto need = [&](const char *name) {
        void *p = dlsym(h, name);
        TORCH_CHECK(p != nullptr, "libsvbrdf_hip.so does not export ", name);
        return p;
    };
    g_abi.mixed = reinterpret_cast<loss_fn_t>(need("svbrdf_mixed_loss_fwd_bwd"));
    g_abi.head = reinterpret_cast<loss_fn_t>(need("svbrdf_head_loss_fwd_bwd"));
    g_abi.mixed_host = reinterpret_cast<loss_fn_t>(need("svbrdf_mixed_loss_fwd_bwd_host_scenes"));
    g_abi.head_host = reinterpret_cast<loss_fn_t>(need("svbrdf_head_loss_fwd_bwd_host_scenes"));
    g_abi.host_rows = reinterpret_cast<int (*)()>(need("svbrdf_host_scenes_max_rows"))();
    g_abi.render_fwd_host = reinterpret_cast<render_fwd_fn_t>(need("svbrdf_render_fwd_host_scenes"));
    g_abi.render_bwd_host = reinterpret_cast<render_bwd_fn_t>(need("svbrdf_render_bwd_host_scenes"));
    g_abi.scale = reinterpret_cast<scale_fn_t>(need("svbrdf_scale_inplace"));
    g_abi.ws_bytes = reinterpret_cast<ws_fn_t>(need("svbrdf_rendering_loss_workspace_bytes"));
    g_abi.last_error = reinterpret_cast<err_fn_t>(need("svbrdf_last_error"));
    g_abi.make_xrow = reinterpret_cast<xrow_fn_t>(need("svbrdf_make_xrow"));
    // the HIP runtime torch already loaded (same SONAME libsvbrdf_hip.so resolved to)
    g_abi.ev_create = reinterpret_cast<ev_create_t>(dlsym(RTLD_DEFAULT, "hipEventCreateWithFlags"));
    g_abi.ev_record = reinterpret_cast<ev_record_t>(dlsym(RTLD_DEFAULT, "hipEventRecord"));
    g_abi.ev_sync = reinterpret_cast<ev_sync_t>(dlsym(RTLD_DEFAULT, "hipEventSynchronize"));
    TORCH_CHECK(g_abi.ev_create && g_abi.ev_record && g_abi.ev_sync, "HIP runtime event symbols not found");
}

void check(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", g_abi.last_error ? g_abi.last_error() : "?");
}

// ------------------------------------------------------------------------------------------
// scene sampler: same draws, same order, same element counts as the reference's per-item loop
// (see svbrdf_estimation_amd/environment.py BatchSceneSampler, which this mirrors op for op)
// ------------------------------------------------------------------------------------------
struct Sampler {
    int64_t B = -1, R = -1, M = -1, D = 0;
    at::Tensor raw, nrm, shift_buf;                 // RNG targets: [B,4R+2M], [B,2,M], [B,M,2]
    at::Tensor r1, phi, radius, cosv, sinv, zin, z, dist;   // flat fp32 work buffers of B*D (dist: B*2*M)
    std::vector<at::Tensor> raw_rows, nrm_rows, nrm_v, nrm_l, shift_rows;
    float lo = 0.0f, width = 0.0f, two_pi = 0.0f, shift_z = 0.0f;

    void init(int64_t b, int64_t r, int64_t m)
    {
        B = b; R = r; M = m; D = 2 * R + M;
        const auto f = at::TensorOptions().dtype(at::kFloat);
        raw = at::empty({B, 4 * R + 2 * M}, f);
        nrm = at::empty({B, 2, M}, f);
        shift_buf = at::empty({B, M, 2}, f);
        for (at::Tensor *t : {&r1, &phi, &radius, &cosv, &sinv, &zin, &z}) *t = at::empty({B * D}, f);
        dist = at::empty({B, 2, M}, f);
        // the constants exactly as torch forms them: float32(0.0 + 0.001), float32(1.0 - 0.1) - lo in fp32,
        // float32(2*pi) (python scalar times fp32 tensor), float32(0.0001) (zeros + 0.0001)
        lo = (float)(0.0 + 0.001);
        width = (float)(1.0 - 0.1) - lo;
        two_pi = (float)(2 * M_PI);
        shift_z = (float)0.0001;
        raw_rows.clear(); nrm_rows.clear(); nrm_v.clear(); nrm_l.clear(); shift_rows.clear();
        for (int64_t i = 0; i < B; ++i) {
            raw_rows.push_back(raw.select(0, i));
            nrm_rows.push_back(nrm.select(0, i));
            nrm_v.push_back(nrm.select(0, i).select(0, 0));
            nrm_l.push_back(nrm.select(0, i).select(0, 1));
            shift_rows.push_back(shift_buf.select(0, i));
        }
    }

    // Only the RNG draws and the five transcendental maps (sqrt, cos, sin, sqrt, exp -- whose CPU
    // implementations are torch's/MKL's and cannot be re-stated bit for bit) go through ATen; the
    // affine steps are plain fp32 loops in the reference's operation order (this file is compiled
    // with -ffp-contract=off; uniform_(lo,hi) is fma(u, hi-lo, lo) of the raw 24-bit u, verified).
    at::Tensor sample()
    {
        auto table = at::empty({B, R + M, 9}, at::TensorOptions().dtype(at::kFloat));
        sample_into(table.data_ptr<float>());
        return table;
    }

    // n draws of Tensor.uniform_(0, 1) on a CPU float tensor, taken straight from the default CPU generator:
    // ATen's kernel is a serial loop of (random() & (2^24-1)) * 2^-24 (ATen/core/TransformationHelper.h
    // uniform_real, ATen/native/cpu/DistributionTemplates.h uniform_kernel) -- the same engine calls in the same
    // order, without 0.4 us of dispatch per call.  `scale`/`shift`: uniform_(lo, hi) is u * (hi - lo) + lo.
    static void draw_uniform(at::CPUGeneratorImpl *gen, float *dst, int64_t n, float scale, float shift)
    {
        std::lock_guard<std::mutex> lock(gen->mutex_);
        for (int64_t i = 0; i < n; ++i) {
            const float u = (float)(gen->random() & 0xFFFFFFu) * (1.0f / 16777216.0f);
            dst[i] = u * scale + shift;     // scale is 1 or 2 here: the product is exact, fused or not
        }
    }

    void sample_into(float *t)      // t: B*(R+M)*9 floats (e.g. a pinned upload slot)
    {
        auto *gen = at::get_generator_or_default<at::CPUGeneratorImpl>(c10::nullopt, at::detail::getDefaultCPUGenerator());
        const int64_t raw_w = 4 * R + 2 * M;
        float *raw_p = raw.data_ptr<float>(), *shift_p = shift_buf.data_ptr<float>(), *nrm_p = nrm.data_ptr<float>();
        for (int64_t i = 0; i < B; ++i) {
            draw_uniform(gen, raw_p + i * raw_w, raw_w, 1.0f, 0.0f);            // == raw_rows[i].uniform_(0, 1)
            if (M > 0) {
                if (2 * M < 16) {
                    // == nrm_v[i].normal_(0.5, 0.75); nrm_l[i].normal_(0.5, 0.75): below 16 elements ATen's CPU kernel is
                    // a serial loop of at::normal_distribution<double> (ATen/native/cpu/DistributionTemplates.h
                    // normal_kernel; the Box-Muller pair is cached in the generator), called here without the dispatch
                    std::lock_guard<std::mutex> lock(gen->mutex_);
                    float *np_ = nrm_p + i * 2 * M;
                    for (int64_t j = 0; j < 2 * M; ++j) {
                        at::normal_distribution<double> normal(0.5, 0.75);
                        np_[j] = static_cast<float>(normal(gen));
                    }
                } else {
                    nrm_v[i].normal_(0.5, 0.75);
                    nrm_l[i].normal_(0.5, 0.75);
                }
                draw_uniform(gen, shift_p + i * M * 2, M * 2, 2.0f, -1.0f);     // == shift_rows[i].uniform_(-1, 1)
            }
        }
        const int64_t W = 4 * R + 2 * M;
        const float *u = raw.data_ptr<float>();
        float *pr1 = r1.data_ptr<float>(), *pphi = phi.data_ptr<float>();
        for (int64_t b = 0; b < B; ++b) {
            const float *row = u + b * W;
            float *o1 = pr1 + b * D, *o2 = pphi + b * D;
            for (int64_t j = 0; j < D; ++j) {           // direction j: view_r | light_r | view_s
                const int64_t i1 = j < R ? j : (j < 2 * R ? 2 * R + (j - R) : 4 * R + (j - 2 * R));
                const int64_t i2 = j < R ? R + j : (j < 2 * R ? 3 * R + (j - R) : 4 * R + M + (j - 2 * R));
                o1[j] = std::fmaf(row[i1], width, lo);
                o2[j] = row[i2] * two_pi;
            }
        }
        at::sqrt_out(radius, r1);
        at::cos_out(cosv, phi);
        at::sin_out(sinv, phi);
        const float *rad = radius.data_ptr<float>();
        float *pz = zin.data_ptr<float>();
        for (int64_t k = 0; k < B * D; ++k) pz[k] = 1.0f - rad[k] * rad[k];
        at::sqrt_out(z, zin);
        if (M > 0) at::exp_out(dist, nrm);
        const float *c = cosv.data_ptr<float>(), *s = sinv.data_ptr<float>(), *zz = z.data_ptr<float>();
        const float *dd = dist.data_ptr<float>(), *sh = shift_buf.data_ptr<float>();
        for (int64_t b = 0; b < B; ++b) {
            float *tb = t + b * (R + M) * 9;
            const int64_t base = b * D;
            for (int64_t j = 0; j < R; ++j) {
                const int64_t v = base + j, l = base + R + j;
                float *o = tb + j * 9;
                o[0] = rad[v] * c[v]; o[1] = rad[v] * s[v]; o[2] = zz[v];
                o[3] = rad[l] * c[l]; o[4] = rad[l] * s[l]; o[5] = zz[l];
                o[6] = o[7] = o[8] = 20.0f;
            }
            for (int64_t m = 0; m < M; ++m) {
                const int64_t v = base + 2 * R + m;
                const float vx = rad[v] * c[v], vy = rad[v] * s[v], vz = zz[v];
                const float dv = dd[(b * 2 + 0) * M + m], dl = dd[(b * 2 + 1) * M + m];
                const float sx = sh[(b * M + m) * 2 + 0], sy = sh[(b * M + m) * 2 + 1];
                float *o = tb + (R + m) * 9;
                o[0] = vx * dv + sx; o[1] = vy * dv + sy; o[2] = vz * dv + shift_z;
                o[3] = (vx * -1.0f) * dl + sx; o[4] = (vy * -1.0f) * dl + sy; o[5] = (vz * 1.0f) * dl + shift_z;
                o[6] = o[7] = o[8] = 50.0f;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
// pinned ring for the table upload (a slot is reused only after its upload's event completed)
// ------------------------------------------------------------------------------------------
struct Ring {
    static constexpr int kDepth = 8;
    at::Tensor slot[kDepth];
    void *event[kDepth] = {nullptr};
    bool used[kDepth] = {false};
    int next = 0;

    // a pinned [numel] fp32 slot that no pending upload still reads
    at::Tensor acquire(int64_t numel, int &index)
    {
        index = next;
        next = (next + 1) % kDepth;
        if (used[index]) g_abi.ev_sync(event[index]);
        if (!slot[index].defined() || slot[index].numel() < numel)
            slot[index] = at::empty({std::max<int64_t>(numel, 1024)}, at::TensorOptions().dtype(at::kFloat).pinned_memory(true));
        return slot[index].slice(0, 0, numel);
    }

    // asynchronous H2D of an acquired slot; the slot is busy until the recorded event completes
    at::Tensor submit(const at::Tensor &view, int index, at::IntArrayRef shape, const at::Device &device, void *stream)
    {
        auto dev = view.view(shape).to(device, /*non_blocking=*/true);
        if (!event[index]) TORCH_CHECK(g_abi.ev_create(&event[index], 0x2 /* hipEventDisableTiming */) == 0, "hipEventCreate failed");
        TORCH_CHECK(g_abi.ev_record(event[index], stream) == 0, "hipEventRecord failed");
        used[index] = true;
        return dev;
    }
};

struct State {
    std::mutex mu;
    void *ev_begin = nullptr, *ev_end = nullptr;   // measurement aid: raw hipEvent_t pair around the next launch
    Sampler sampler;
    Ring ring;
    // one device per process (one process per GPU); scratch per stream so that calls enqueued on
    // different streams never share the loss accumulators; xrow per width
    std::map<int64_t, at::Tensor> workspace_by_stream;
    std::map<int64_t, at::Tensor> xrow_by_width;
    at::Tensor workspace, xrow;      // the ones selected for the call in flight (under `mu`)
    at::Tensor host_table;           // [B,S,9] sampler target of the by-value route (consumed inside the call)
    int device = -1;
    // the device-resident upstream gradient 1.0 that `loss.backward()` is given instead of letting the engine fill a fresh
    // ones tensor (unit_gradient below); the node recognises it by address and skips its scale launch
    // It is handed to Python (and through the engine to any hook on the loss), so "this is 1.0" is only believed while the
    // tensor's version counter still reads what it read at creation: an in-place edit (a loss-scaling hook doing
    // g.mul_(k)) keeps the address but bumps the version -- the node then applies the scale like for any other gradient,
    // and the next backward gets a fresh 1.0 (ensure_unit_grad).  Created with the workspace at the first forward on a
    // device, i.e. outside any later stream capture.
    at::Tensor unit_grad;
    std::atomic<const void *> unit_ptr{nullptr};
    std::atomic<uint32_t> unit_version{0};

    // under `mu`
    void ensure_unit_grad(const at::TensorOptions &like)
    {
        if (unit_grad.defined() && unit_grad.device() == like.device() && unit_grad._version() == unit_version.load(std::memory_order_relaxed))
            return;
        unit_ptr.store(nullptr, std::memory_order_relaxed);
        unit_grad = at::ones({}, like.dtype(at::kFloat).requires_grad(false));
        unit_version.store(unit_grad._version(), std::memory_order_relaxed);
        unit_ptr.store(unit_grad.data_ptr(), std::memory_order_relaxed);
    }
};
// Deliberately never destroyed: the state owns device tensors, pinned host slots and events, and a static
// destructor would release them AFTER the HIP runtime has shut down at interpreter exit (observed: a process that
// had used the pinned upload ring printed its result and then hung in teardown).
State &g_state = *new State();

// Second order.  backward(create_graph=True) runs a node's apply() with grad mode ON and expects a gradient that is itself
// attached to the graph; the kernels' gradient buffers are constants to autograd.  The nodes below then hand the call to
// the package's Python side (losses.differentiable_loss_backward, renderers.differentiable_render_backward: the same loss
// composed from the float64 render kernels, whose backward has a dual-number companion kernel), registered here at import.
// Never destroyed (like g_state): Python objects must not be released at static-destruction time.
struct SecondOrderHooks {
    pybind11::object loss, render;
};
SecondOrderHooks &g_hooks = *new SecondOrderHooks();

// ------------------------------------------------------------------------------------------
// autograd node: the kernel has already produced d loss / d input for upstream gradient 1.
// A plain torch::autograd::Node (not the Function<> template: no AutogradContext, no saved_data dictionary of
// IValues, no variable wrapping pass -- about 3 us less host time per step).
// ------------------------------------------------------------------------------------------
struct FusedLossBackward : public torch::autograd::Node {
    at::Tensor grad_in, grad_tg;     // moved out by apply()
    bool has_in = false, has_tg = false, done = false;
    void *stream = nullptr;          // backward is enqueued on the forward's stream
    // for backward(create_graph=True) only: the inputs (version-checked), the scene table and the loss's parameters
    torch::autograd::SavedVariable input, target;
    at::Tensor scenes_dev;
    std::vector<float> scenes_host;
    int64_t B = 0, S = 0;
    double eps = 0.0, l1_weight = 0.0, eps_l1 = 0.0;
    bool head = false;

    torch::autograd::variable_list second_order(const at::Tensor &g0)
    {
        pybind11::gil_scoped_acquire gil;
        TORCH_CHECK(g_hooks.loss && !g_hooks.loss.is_none(), "fused rendering loss: backward(create_graph=True) needs the "
                                                               "package's second-order hook (import svbrdf_estimation_amd.losses)");
        const auto in = input.unpack(shared_from_this()), tg = target.unpack(shared_from_this());
        TORCH_CHECK(in.defined() && tg.defined(), "Trying to backward through the fused rendering loss a second time: its "
                                                  "saved tensors were freed; specify retain_graph=True");
        const at::Tensor sc = scenes_dev.defined()
                                  ? scenes_dev
                                  : at::from_blob(scenes_host.data(), {B, S, 9}, at::TensorOptions().dtype(at::kFloat)).clone();
        torch::autograd::variable_list out(2);
        try {
            const pybind11::tuple r = g_hooks.loss(in, tg, sc, eps, l1_weight, eps_l1, head, g0, has_in, has_tg);
            if (!r[0].is_none()) out[0] = r[0].cast<at::Tensor>();
            if (!r[1].is_none()) out[1] = r[1].cast<at::Tensor>();
        } catch (pybind11::error_already_set &e) {
            // an autograd worker thread: hand the engine a plain C++ exception; the Python error state dies here, under the GIL
            const std::string msg = e.what();
            throw std::runtime_error("fused rendering loss, backward(create_graph=True): " + msg);
        }
        return out;
    }

    torch::autograd::variable_list apply(torch::autograd::variable_list &&grads) override
    {
        if (at::GradMode::is_enabled()) {        // backward(create_graph=True)
            TORCH_CHECK(grads[0].defined(), "fused rendering loss: undefined upstream gradient");
            return second_order(grads[0]);
        }
        TORCH_CHECK(!done, "Trying to backward through the fused rendering loss a second time: its gradient buffers were "
                           "handed to the first backward.  Specify retain_graph=True for that call (the buffers then stay "
                           "with the graph and each backward receives a scaled copy), as with any autograd graph "
                           "(losses.py:29-52 is plain autograd in the reference).");
        // the upstream gradient of the scalar loss, as a one-element fp32 device buffer (loss.backward() hands over
        // exactly that: no ops needed)
        const at::Tensor &g0 = grads[0];
        TORCH_CHECK(g0.defined(), "fused rendering loss: undefined upstream gradient");
        const auto scale = (g0.scalar_type() == at::kFloat && g0.numel() == 1 && g0.is_contiguous())
                               ? g0 : g0.detach().to(at::kFloat).reshape({1});
        // retain_graph=True: the node must stay usable, so the kernel's buffers stay here and the caller gets a scaled
        // COPY (one extra pass over the gradient).  Otherwise -- the training loop -- the buffers are MOVED out and scaled
        // in place: AccumulateGrad adopts a gradient it holds the only reference to and clones it otherwise (measured:
        // a 25 MB device copy, 7.4 us, per step).
        const bool keep = torch::autograd::get_current_graph_task_keep_graph();
        if (!keep) done = true;
        // A plain `loss.backward()` arrives here with THE unit gradient (losses._FusedLossTensor.backward hands the engine
        // the cached device-resident 1.0 of unit_gradient() instead of letting it fill a fresh ones tensor): recognised by
        // address, it needs no scaling at all -- the step is then ONE kernel launch, as on the engine-free leaf path.
        // (address AND version counter: a hook that edited the tensor in place keeps the address; then it is a scale like any)
        const bool unit = g0.is_cuda() && g0.data_ptr() == g_state.unit_ptr.load(std::memory_order_relaxed) &&
                          g0._version() == g_state.unit_version.load(std::memory_order_relaxed);
        torch::autograd::variable_list out(2);
        at::AutoDispatchBelowADInplaceOrView below_autograd;
        if (has_in) {
            out[0] = keep ? grad_in.clone() : std::move(grad_in);
            if (!unit) check(g_abi.scale(out[0].data_ptr<float>(), scale.data_ptr<float>(), (size_t)out[0].numel(), stream), "svbrdf_scale_inplace");
        }
        if (has_tg) {
            out[1] = keep ? grad_tg.clone() : std::move(grad_tg);
            if (!unit) check(g_abi.scale(out[1].data_ptr<float>(), scale.data_ptr<float>(), (size_t)out[1].numel(), stream), "svbrdf_scale_inplace");
        }
        return out;
    }

    void release_variables() override
    {
        done = true;
        grad_in.reset();
        grad_tg.reset();
        input.reset_data();
        target.reset_data();
        scenes_dev.reset();
    }

    std::string name() const override { return "SvbrdfFusedLossBackward"; }
};

// launches the fused kernel(s) and returns the 0-dim loss with the node attached
at::Tensor run_fused(const at::Tensor &input, const at::Tensor &target, const at::Tensor &scenes, double eps, double l1_weight,
                     double eps_l1, int64_t stream, bool head)
{
    const bool grad_mode = at::GradMode::is_enabled();
    const bool need_in = grad_mode && input.requires_grad(), need_tg = grad_mode && target.requires_grad();
    TORCH_CHECK(!(head && need_tg), "the head-fused loss has no gradient w.r.t. the target maps");
    // a host table goes into the kernel-argument block of the launch, a device table is read in place
    const bool host_table = scenes.is_cpu();
    const loss_fn_t kernel = host_table ? (head ? g_abi.head_host : g_abi.mixed_host) : (head ? g_abi.head : g_abi.mixed);
    const loss_fn_t kernel_tg = host_table ? g_abi.mixed_host : g_abi.mixed;
    at::Tensor loss, grad_in, grad_tg;
    void *st = reinterpret_cast<void *>(stream);
    {
        at::AutoDispatchBelowADInplaceOrView below_autograd;    // plain buffers: nothing here is recorded
        const auto in = input.contiguous(), tg = target.contiguous();
        const int B = (int)in.size(0), S = (int)scenes.size(1), H = (int)in.size(2), W = (int)in.size(3);
        loss = at::empty({}, in.options());
        const size_t ws_bytes = (size_t)g_state.workspace.numel() * 8;
        if (need_in) grad_in = at::empty_like(in);
        if (g_state.ev_begin) g_abi.ev_record(g_state.ev_begin, st);
        const int rc = kernel(in.data_ptr<float>(), tg.data_ptr<float>(), scenes.data_ptr<float>(),
                              g_state.xrow.data_ptr<float>(), (float)eps, (float)l1_weight, (float)eps_l1,
                              loss.data_ptr<float>(), need_in ? grad_in.data_ptr<float>() : nullptr,
                              g_state.workspace.data_ptr(), ws_bytes, B, S, H, W, st);
        if (g_state.ev_end) g_abi.ev_record(g_state.ev_end, st);
        g_state.ev_begin = g_state.ev_end = nullptr;
        if (rc != 0) g_state.workspace.zero_();   // only completed calls leave the scratch zeroed (svbrdf_hip.h)
        check(rc, "svbrdf_mixed_loss_fwd_bwd");
        if (need_tg) {   // every term is |g(a) - g(b)|: the target's gradient is the same kernel, roles swapped
            grad_tg = at::empty_like(tg);
            auto loss2 = at::empty({1}, in.options());
            const int rc2 = kernel_tg(tg.data_ptr<float>(), in.data_ptr<float>(), scenes.data_ptr<float>(),
                                      g_state.xrow.data_ptr<float>(), (float)eps, (float)l1_weight, (float)eps_l1,
                                      loss2.data_ptr<float>(), grad_tg.data_ptr<float>(), g_state.workspace.data_ptr(),
                                      ws_bytes, B, S, H, W, st);
            if (rc2 != 0) g_state.workspace.zero_();
            check(rc2, "svbrdf_mixed_loss_fwd_bwd (target)");
        }
    }
    if (need_in || need_tg) {
        auto node = std::shared_ptr<FusedLossBackward>(new FusedLossBackward(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(input, target));
        node->grad_in = std::move(grad_in);
        node->grad_tg = std::move(grad_tg);
        node->has_in = need_in;
        node->has_tg = need_tg;
        node->stream = st;
        node->input = torch::autograd::SavedVariable(input, false);
        node->target = torch::autograd::SavedVariable(target, false);
        node->B = scenes.size(0);
        node->S = scenes.size(1);
        if (host_table) node->scenes_host.assign(scenes.data_ptr<float>(), scenes.data_ptr<float>() + scenes.numel());
        else node->scenes_dev = scenes;
        node->eps = eps; node->l1_weight = l1_weight; node->eps_l1 = eps_l1; node->head = head;
        torch::autograd::set_history(loss, node);
    }
    return loss;
}

void ensure_device_state(const at::Tensor &input, int S, int64_t stream)
{
    const int B = (int)input.size(0), H = (int)input.size(2), W = (int)input.size(3);
    const int dev = input.device().index();
    if (g_state.device != dev) {          // first call, or the process switched devices: drop everything cached
        g_state.workspace_by_stream.clear();
        g_state.xrow_by_width.clear();
        g_state.device = dev;
    }
    const size_t need = g_abi.ws_bytes(B, S, H, W);
    auto &ws = g_state.workspace_by_stream[stream];
    if (!ws.defined() || (size_t)ws.numel() * 8 < need)
        ws = at::zeros({(int64_t)std::max<size_t>((need + 7) / 8, 8)}, at::TensorOptions().dtype(at::kLong).device(input.device()));
    g_state.workspace = ws;
    auto &xr = g_state.xrow_by_width[W];
    if (!xr.defined()) {
        auto host = at::empty({W}, at::TensorOptions().dtype(at::kFloat));
        check(g_abi.make_xrow(host.data_ptr<float>(), W), "svbrdf_make_xrow");
        xr = host.to(input.device());
    }
    g_state.xrow = xr;
    g_state.ensure_unit_grad(input.options());
}

void check_inputs(const at::Tensor &input, const at::Tensor &target, bool head)
{
    TORCH_CHECK(g_abi.mixed != nullptr, "host extension not bound to libsvbrdf_hip.so (call bind first)");
    TORCH_CHECK(input.dim() == 4 && target.dim() == 4 && target.size(1) == 12 && input.size(1) == (head ? 9 : 12) &&
                    input.size(0) == target.size(0) && input.size(2) == target.size(2) && input.size(3) == target.size(3),
                head ? "input must be [B,9,H,W] and target [B,12,H,W]" : "input and target must both be [B,12,H,W]");
    TORCH_CHECK(input.is_cuda() && target.is_cuda() && input.device() == target.device(),
                "the MI355X engine only computes on a ROCm device (no CPU fallback)");
    TORCH_CHECK(input.scalar_type() == at::kFloat && target.scalar_type() == at::kFloat, "fp32 only");
    TORCH_CHECK(input.size(2) == input.size(3), "H must equal W");
}

}  // namespace

// scenes drawn here, reference RNG order; returns the 0-dim loss on the device
at::Tensor fused_loss(const at::Tensor &input, const at::Tensor &target, int64_t n_random, int64_t n_specular,
                      double eps, double l1_weight, double eps_l1, int64_t stream, bool head)
{
    check_inputs(input, target, head);
    std::lock_guard<std::mutex> lock(g_state.mu);
    const int64_t B = input.size(0);
    if (g_state.sampler.B != B || g_state.sampler.R != n_random || g_state.sampler.M != n_specular)
        g_state.sampler.init(B, n_random, n_specular);
    const int64_t S = n_random + n_specular;
    ensure_device_state(input, (int)S, stream);
    if (B * S <= g_abi.host_rows) {
        // small table (B*S <= 288: every BASELINE configuration at its per-GPU batch): drawn into a host buffer and handed to the
        // launch by value -- the step is ONE dispatch, no H2D command, no device buffer, no pinned slot
        if (!g_state.host_table.defined() || g_state.host_table.size(0) != B || g_state.host_table.size(1) != S)
            g_state.host_table = at::empty({B, S, 9}, at::TensorOptions().dtype(at::kFloat));
        g_state.sampler.sample_into(g_state.host_table.data_ptr<float>());
        return run_fused(input, target, g_state.host_table, eps, l1_weight, eps_l1, stream, head);
    }
    int slot = 0;
    auto pinned = g_state.ring.acquire(B * S * 9, slot);
    g_state.sampler.sample_into(pinned.data_ptr<float>());           // drawn straight into the upload slot
    const auto scenes = g_state.ring.submit(pinned, slot, {B, S, 9}, input.device(), reinterpret_cast<void *>(stream));
    return run_fused(input, target, scenes, eps, l1_weight, eps_l1, stream, head);
}

// same with caller-provided scenes ([B,S,9] on the device, or on the host if B*S <= svbrdf_host_scenes_max_rows())
at::Tensor fused_loss_with_scenes(const at::Tensor &input, const at::Tensor &target, const at::Tensor &scenes, double eps,
                                  double l1_weight, double eps_l1, int64_t stream, bool head)
{
    check_inputs(input, target, head);
    TORCH_CHECK(scenes.dim() == 3 && scenes.size(0) == input.size(0) && scenes.size(2) == 9 &&
                    scenes.scalar_type() == at::kFloat,
                "scenes must be a [B,S,9] fp32 tensor");
    TORCH_CHECK(scenes.is_cuda() || scenes.size(0) * scenes.size(1) <= g_abi.host_rows,
                "a host scene table may hold at most ", g_abi.host_rows, " rows; upload larger ones first");
    std::lock_guard<std::mutex> lock(g_state.mu);
    ensure_device_state(input, (int)scenes.size(1), stream);
    return run_fused(input, target, scenes.contiguous(), eps, l1_weight, eps_l1, stream, head);
}

// ------------------------------------------------------------------------------------------
// LocalRenderer.render(scene, svbrdf) without the interpreter in the loop: K1 with the scene's rows by value, and a
// plain autograd node whose apply() launches K2.  `rows` is the HOST [S,9] table shared by every map of the batch
// (renderers.py:98: one scene, B maps); it is copied into the node (9*S floats), nothing of the caller's is retained.
// ------------------------------------------------------------------------------------------
struct RenderBackward : public torch::autograd::Node {
    torch::autograd::SavedVariable maps;      // version-checked like any saved input
    std::vector<float> rows;
    at::Tensor xrow;
    int B = 0, S = 0, H = 0, W = 0;
    void *stream = nullptr;

    torch::autograd::variable_list apply(torch::autograd::variable_list &&grads) override
    {
        torch::autograd::variable_list out(1);
        if (!task_should_compute_output(0)) return out;
        const auto m = maps.unpack(shared_from_this());
        TORCH_CHECK(m.defined(), "Trying to backward through LocalRenderer.render a second time: the saved maps were freed; "
                                 "specify retain_graph=True for the first backward");
        if (at::GradMode::is_enabled() && grads[0].defined()) {        // backward(create_graph=True): see SecondOrderHooks
            pybind11::gil_scoped_acquire gil;
            TORCH_CHECK(g_hooks.render && !g_hooks.render.is_none(), "LocalRenderer.render: backward(create_graph=True) needs "
                                                                       "the package's second-order hook");
            const at::Tensor table = at::from_blob(rows.data(), {(int64_t)S, 9}, at::TensorOptions().dtype(at::kFloat)).clone();
            try {
                out[0] = g_hooks.render(m, table, grads[0]).cast<at::Tensor>();
            } catch (pybind11::error_already_set &e) {
                const std::string msg = e.what();
                    throw std::runtime_error("LocalRenderer.render, backward(create_graph=True): " + msg);
            }
            return out;
        }
        at::AutoDispatchBelowADInplaceOrView below_autograd;
        if (!grads[0].defined()) {               // an undefined cotangent means zeros
            out[0] = at::zeros_like(m);
            return out;
        }
        const auto go = grads[0].contiguous();
        TORCH_CHECK(go.scalar_type() == at::kFloat && go.numel() == (int64_t)B * S * 3 * H * W, "render backward: bad cotangent");
        auto grad = at::empty_like(m);
        check(g_abi.render_bwd_host(m.data_ptr<float>(), rows.data(), 1, xrow.data_ptr<float>(), go.data_ptr<float>(),
                                    grad.data_ptr<float>(), B, S, H, W, stream), "svbrdf_render_bwd_host_scenes");
        out[0] = std::move(grad);
        return out;
    }

    void release_variables() override { maps.reset_data(); }

    std::string name() const override { return "SvbrdfRenderBackward"; }
};

// maps [B,12,H,W] on the device, rows: HOST fp32 [S,9] shared by all maps -> [B,S,3,H,W]
at::Tensor render_shared_scenes(const at::Tensor &maps_in, const at::Tensor &rows, int64_t stream)
{
    TORCH_CHECK(g_abi.render_fwd_host != nullptr, "host extension not bound to libsvbrdf_hip.so (call bind first)");
    TORCH_CHECK(maps_in.dim() == 4 && maps_in.size(1) == 12 && maps_in.size(2) == maps_in.size(3),
                "svbrdf must be [B,12,H,W] with H == W");
    TORCH_CHECK(maps_in.is_cuda() && maps_in.scalar_type() == at::kFloat,
                "the MI355X engine only computes on fp32 tensors on a ROCm device (no CPU fallback)");
    TORCH_CHECK(rows.is_cpu() && rows.scalar_type() == at::kFloat && rows.dim() == 2 && rows.size(1) == 9 &&
                    rows.is_contiguous() && rows.size(0) >= 1 && rows.size(0) <= g_abi.host_rows,
                "rows must be a contiguous host fp32 [S,9] table of at most ", g_abi.host_rows, " scenes");
    const int B = (int)maps_in.size(0), S = (int)rows.size(0), H = (int)maps_in.size(2), W = (int)maps_in.size(3);
    at::Tensor xrow;
    {
        std::lock_guard<std::mutex> lock(g_state.mu);
        const int dev = maps_in.device().index();
        if (g_state.device != dev) {
            g_state.workspace_by_stream.clear();
            g_state.xrow_by_width.clear();
            g_state.device = dev;
        }
        auto &xr = g_state.xrow_by_width[W];
        if (!xr.defined()) {
            auto host = at::empty({W}, at::TensorOptions().dtype(at::kFloat));
            check(g_abi.make_xrow(host.data_ptr<float>(), W), "svbrdf_make_xrow");
            xr = host.to(maps_in.device());
        }
        xrow = xr;
    }
    void *st = reinterpret_cast<void *>(stream);
    at::Tensor out, maps;
    {
        at::AutoDispatchBelowADInplaceOrView below_autograd;
        maps = maps_in.contiguous();
        out = at::empty({B, S, 3, H, W}, maps.options());
        check(g_abi.render_fwd_host(maps.data_ptr<float>(), rows.data_ptr<float>(), 1, xrow.data_ptr<float>(),
                                    out.data_ptr<float>(), B, S, H, W, st), "svbrdf_render_fwd_host_scenes");
    }
    if (at::GradMode::is_enabled() && maps_in.requires_grad()) {
        auto node = std::shared_ptr<RenderBackward>(new RenderBackward(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(maps_in));
        node->maps = torch::autograd::SavedVariable(maps_in.is_contiguous() ? maps_in : maps, false);
        node->rows.assign(rows.data_ptr<float>(), rows.data_ptr<float>() + (size_t)S * 9);
        node->xrow = xrow;
        node->B = B; node->S = S; node->H = H; node->W = W;
        node->stream = st;
        torch::autograd::set_history(out, node);
    }
    return out;
}

// The upstream gradient of a plain `loss.backward()`: a 0-dim float32 1.0 on the loss's device, created once per device
// (with the workspace, at the first forward) and never written by this extension; callers must not write it either -- if
// one does, the version counter gives it away (State::ensure_unit_grad).  Handing it to torch.autograd.backward as the explicit gradient saves the engine's fill kernel, and
// FusedLossBackward::apply recognises it by address and skips its own (no-op) scale launch.
at::Tensor unit_gradient(const at::Tensor &loss)
{
    TORCH_CHECK(loss.is_cuda() && loss.scalar_type() == at::kFloat, "unit_gradient: the loss must be a float32 device tensor");
    std::lock_guard<std::mutex> lock(g_state.mu);
    g_state.ensure_unit_grad(loss.options());      // normally a no-op (made with the workspace); a mutated one is replaced
    return g_state.unit_grad;
}

// LocalRenderer.render(scene, svbrdf) with the scene's nine numbers as plain Python floats (camera xyz | light xyz | rgb): the
// one-scene call of the plugin interface without building a tensor for the row on the Python side (2.5 us of an 11 us call).
// Rounded to float32 here exactly as torch.Tensor([...]) rounds them (renderers.py:79,91,98).
at::Tensor render_one_scene(const at::Tensor &maps_in, const std::vector<double> &scene9, int64_t stream)
{
    TORCH_CHECK(scene9.size() == 9, "a scene is nine numbers: camera xyz, light xyz, light rgb");
    auto rows = at::empty({1, 9}, at::TensorOptions().dtype(at::kFloat));
    float *r = rows.data_ptr<float>();
    for (int i = 0; i < 9; ++i) r[i] = static_cast<float>(scene9[i]);
    return render_shared_scenes(maps_in, rows, stream);
}

// A plain `loss.backward()` (no explicit gradient, no create_graph, no inputs=) entered from C++: PyTorch's autograd engine
// runs the graph below `loss` exactly as for torch.autograd.backward(loss, unit_gradient) -- same engine, same nodes, same
// hooks -- through the public C++ entry point torch::autograd::backward, without the Python argument processing in front
// of it (tensor -> tuple conversions, _make_grads' shape checks, the _engine_run_backward wrapper: ~6 us of interpreter
// time per step, which counts when the whole step is one 36 us kernel).  The GIL is released while the engine runs, as
// THPEngine_run_backward does; Python-defined nodes and hooks of the graph re-acquire it themselves.
void engine_backward(const at::Tensor &loss, bool retain_graph)
{
    TORCH_CHECK(loss.is_cuda() && loss.scalar_type() == at::kFloat && loss.numel() == 1,
                "engine_backward: the loss must be a one-element float32 device tensor");
    at::Tensor g;
    {
        std::lock_guard<std::mutex> lock(g_state.mu);
        g_state.ensure_unit_grad(loss.options());
        g = g_state.unit_grad;
    }
    pybind11::gil_scoped_release no_gil;
    torch::autograd::backward({loss}, {g}, retain_graph, /*create_graph=*/false, /*inputs=*/{});
}

// ------------------------------------------------------------------------------------------
// input-photo scenes (row f3): the draws of SvbrdfDataset.render_inputs (dataset.py:172-204) for B samples of n photos each,
// sample after sample in the reference's call order -- synthesis.input_scene_table restated without ~14 small tensor ops
// per sample (25 us of interpreter time each; a batch of 8 was host-bound at 0.2 ms per call).  Per sample:
//   uniform_(-0.75, 0.75) x2                    light x, y of the fronto-parallel first photo (z = 2.197)
//   [n > 1]  direction sampler, n-1 lights      r1 ~ U(0.001, 0.98) then r2 ~ U(0, 1) (utils.py:100-111), x 2.197
//   [aug]    normal_(-2, 0.5) x1 -> exp         spread of the light power;  normal_(20, spread) x n -> abs
//   [aug]    normal_(1, 0.03) x 3n -> abs       white balance;  uniform_(0.25, 2.75) x n   view distances
//   uniform_(-0.25, 0.25) x2                    view x, y of the first photo
//   [n > 1]  direction sampler, n-1 views       x the photo's view distance
// Uniform draws and normal draws of fewer than 16 elements are taken straight from the default CPU generator the way ATen's
// serial kernels take them (see Sampler::draw_uniform; uniform_(lo, hi) = fma(u, hi - lo, lo) with hi - lo formed in
// float); 16 and more normals go through Tensor.normal_ (ATen switches to its vectorised fill there).  sqrt / cos / sin / exp
// are ATen's.  tests/test_host_logic.py: tables AND generator state bit-identical to the per-sample Python functions.
// ------------------------------------------------------------------------------------------
namespace {
void draw_uniform_range(at::CPUGeneratorImpl *gen, float *dst, int64_t n, float lo, float hi)
{
    const float width = hi - lo;
    std::lock_guard<std::mutex> lock(gen->mutex_);
    for (int64_t i = 0; i < n; ++i)
        dst[i] = std::fmaf((float)(gen->random() & 0xFFFFFFu) * (1.0f / 16777216.0f), width, lo);
}

// Tensor(n).normal_(mean, std) into dst
void draw_normal(at::CPUGeneratorImpl *gen, float *dst, int64_t n, double mean, double stdv)
{
    if (n >= 16) {
        auto t = at::from_blob(dst, {n}, at::TensorOptions().dtype(at::kFloat));
        t.normal_(mean, stdv);
        return;
    }
    std::lock_guard<std::mutex> lock(gen->mutex_);
    for (int64_t i = 0; i < n; ++i) {
        at::normal_distribution<double> normal(mean, stdv);
        dst[i] = static_cast<float>(normal(gen));
    }
}
}  // namespace

at::Tensor sample_input_scene_table(int64_t B, int64_t n, bool augment)
{
    TORCH_CHECK(B >= 1 && n >= 1, "sample_input_scene_table: B and the photo count must be positive");
    auto *gen = at::get_generator_or_default<at::CPUGeneratorImpl>(c10::nullopt, at::detail::getDefaultCPUGenerator());
    const auto f = at::TensorOptions().dtype(at::kFloat);
    const int64_t d = n - 1;                                   // hemisphere directions per list
    const float light_z = (float)2.197, fixed_view = (float)2.75, two_pi = (float)(2 * M_PI);
    const float dir_lo = (float)(0.0 + 0.001), dir_hi = (float)(1.0 - 0.02);
    auto table = at::empty({B, n, 9}, f);
    // raw draws: [b][0 = lights, 1 = views][d]
    auto r1 = at::empty({B, 2, std::max<int64_t>(d, 1)}, f), r2 = at::empty_like(r1);
    std::vector<float> light_xy(2 * B), view_xy(2 * B), vdist(B * n, fixed_view), colour(B * n * 3, 30.0f), power(n), wb(3 * n);
    auto spread = at::empty({1}, f);
    float *p1 = r1.data_ptr<float>(), *p2 = r2.data_ptr<float>();
    for (int64_t b = 0; b < B; ++b) {
        draw_uniform_range(gen, &light_xy[2 * b], 2, -0.75f, 0.75f);
        if (d > 0) {
            draw_uniform_range(gen, p1 + (b * 2 + 0) * d, d, dir_lo, dir_hi);
            draw_uniform_range(gen, p2 + (b * 2 + 0) * d, d, 0.0f, 1.0f);
        }
        if (augment) {
            draw_normal(gen, spread.data_ptr<float>(), 1, -2.0, 0.5);
            const float sd = at::exp(spread).data_ptr<float>()[0];          // torch.exp(...).numpy()[0]
            draw_normal(gen, power.data(), n, 20.0, (double)sd);
            draw_normal(gen, wb.data(), 3 * n, 1.0, 0.03);
            for (int64_t i = 0; i < n; ++i)
                for (int c = 0; c < 3; ++c) colour[(b * n + i) * 3 + c] = std::fabs(power[i]) * std::fabs(wb[3 * i + c]);
            draw_uniform_range(gen, &vdist[b * n], n, 0.25f, 2.75f);
        }
        draw_uniform_range(gen, &view_xy[2 * b], 2, -0.25f, 0.25f);
        if (d > 0) {
            draw_uniform_range(gen, p1 + (b * 2 + 1) * d, d, dir_lo, dir_hi);
            draw_uniform_range(gen, p2 + (b * 2 + 1) * d, d, 0.0f, 1.0f);
        }
    }
    at::Tensor radius, cosv, sinv, z;
    if (d > 0) {                                               // utils.generate_normalized_random_direction, whole batch at once
        radius = at::sqrt(r1);
        auto phi = r2 * two_pi;
        cosv = at::cos(phi);
        sinv = at::sin(phi);
        z = at::sqrt(1.0f - radius * radius);
    }
    float *t = table.data_ptr<float>();
    for (int64_t b = 0; b < B; ++b) {
        float *row = t + b * n * 9;
        row[0] = view_xy[2 * b]; row[1] = view_xy[2 * b + 1]; row[2] = vdist[b * n];
        row[3] = light_xy[2 * b]; row[4] = light_xy[2 * b + 1]; row[5] = light_z;
        for (int64_t i = 1; i < n; ++i) {
            const int64_t l = (b * 2 + 0) * d + (i - 1), v = (b * 2 + 1) * d + (i - 1);
            const float *rad = radius.data_ptr<float>(), *c = cosv.data_ptr<float>(), *sn = sinv.data_ptr<float>(),
                        *zz = z.data_ptr<float>();
            float *o = row + i * 9;
            const float vd = vdist[b * n + i];
            o[0] = (rad[v] * c[v]) * vd; o[1] = (rad[v] * sn[v]) * vd; o[2] = zz[v] * vd;
            o[3] = (rad[l] * c[l]) * light_z; o[4] = (rad[l] * sn[l]) * light_z; o[5] = zz[l] * light_z;
        }
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) row[i * 9 + 6 + c] = colour[(b * n + i) * 3 + c];
    }
    return table;
}

// the sampler alone (host tensor) -- used by the bit-exactness tests
at::Tensor sample_scene_table(int64_t batch, int64_t n_random, int64_t n_specular)
{
    std::lock_guard<std::mutex> lock(g_state.mu);
    if (g_state.sampler.B != batch || g_state.sampler.R != n_random || g_state.sampler.M != n_specular)
        g_state.sampler.init(batch, n_random, n_specular);
    return g_state.sampler.sample();
}

// measurement aid (bench.py): record this raw hipEvent_t pair around the NEXT kernel launch only
void set_timing_events(int64_t begin, int64_t end)
{
    std::lock_guard<std::mutex> lock(g_state.mu);
    g_state.ev_begin = reinterpret_cast<void *>(begin);
    g_state.ev_end = reinterpret_cast<void *>(end);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("set_timing_events", &set_timing_events);
    m.def("bind", &bind, "resolve the C ABI of libsvbrdf_hip.so (path)");
    m.def("fused_loss", &fused_loss);
    m.def("fused_loss_with_scenes", &fused_loss_with_scenes);
    m.def("sample_scene_table", &sample_scene_table);
    m.def("sample_input_scene_table", &sample_input_scene_table,
          "[B,n,9] scenes of the input-photo synthesis, reference draw order (dataset.py:172-204)");
    m.def("unit_gradient", &unit_gradient);
    m.def("engine_backward", &engine_backward, "loss.backward() through torch::autograd::backward with the unit gradient");
    m.def("render_shared_scenes", &render_shared_scenes);
    m.def("render_one_scene", &render_one_scene);
    m.def("set_second_order_hooks", [](pybind11::object loss, pybind11::object render) {
        g_hooks.loss = std::move(loss);
        g_hooks.render = std::move(render);
    }, "callables the autograd nodes hand backward(create_graph=True) to (losses.differentiable_loss_backward, "
       "renderers.differentiable_render_backward)");
}
