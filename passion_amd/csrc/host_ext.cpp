// Host side of the two hottest ops as C++ autograd nodes: the fused conv block (forward + backward) and the
// InstanceNorm statistics.  A training step issues ~110 conv blocks and ~95 statistics calls in each direction; as Python
// autograd.Functions over ctypes the step was HOST-bound on MI355X (19 ms of interpreter + binding time against < 19 ms of
// kernels): descriptor filling, output / scratch allocation and the backward nodes (which the autograd engine otherwise
// runs under the GIL) live here instead.  The kernels are reached through the same C ABI as everything else
// (include/passion_hip.h); the entry points are handed over as addresses by passion_amd/_lib.py (bind()), so the library
// that ctypes loaded -- and nothing else -- is what runs.
//
// Mirrors passion_amd/ops.py: ConvBlockFn (reference general_conv3d_prenorm, code/models/blocks.py:300-316) and
// instnorm_stats (nn.InstanceNorm3d statistics, blocks.py:18).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include "passion_hip.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using OptT = std::optional<at::Tensor>;

struct Abi {
    decltype(&passion_conv3d_fwd) conv_fwd = nullptr;
    decltype(&passion_conv3d_fwd_stats) conv_fwd_stats = nullptr;
    decltype(&passion_conv3d_fwd_grouped) conv_fwd_grouped = nullptr;
    decltype(&passion_conv3d_bwd_grouped) conv_bwd_grouped = nullptr;
    decltype(&passion_conv3d_bwd) conv_bwd = nullptr;
    decltype(&passion_conv_bwd_scratch_floats) conv_bwd_scratch = nullptr;
    decltype(&passion_instnorm_stats) stats = nullptr;
    decltype(&passion_instnorm_scratch_doubles) stats_scratch = nullptr;
} g;

inline void* cur_stream() { return reinterpret_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

inline int dtype_code(at::ScalarType t) {
    if (t == at::kFloat) return PASSION_F32;
    if (t == at::kBFloat16) return PASSION_BF16;
    TORCH_CHECK(false, "passion_amd: unsupported activation dtype ", t);
}

inline void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed with code ", rc); }

inline void need_gpu(const at::Tensor& t) {
    TORCH_CHECK(t.is_cuda(), "passion_amd kernels need GPU tensors (no CPU fallback); got a ", t.device(), " tensor");
}

// per segment: C, offset (elements), bstride (elements), bmod, mask_col
constexpr int SEG_INTS = 5;
// geo: B, D, H, W, ksize, stride, pad_mode, out dtype code
constexpr int GEO_INTS = 8;

struct ConvCfg {
    std::vector<int64_t> geo, seg;
    double slope;
    std::vector<OptT> stats, masks;
    at::Tensor packed;
    OptT out;
    bool want_stats = false;     // also return the InstanceNorm statistics of y (fused into the conv kernel where it can be)
    OptT stats_out;              // ... written into this preallocated (B, Cout, 2) f32 slot
    double eps = 1e-5;
    std::vector<at::Tensor> params;     // direct-gradient mode: the parameters (weights, then biases) whose .grad the backward fills itself
    int pass_x = 0;                     // grouped node: hand xs[0] on as 1 or 2 further OUTPUTS of the node (below)
};

// Direct-gradient mode (set_direct_grads, single-process training): the conv nodes take DETACHED weights / biases and write the
// parameters' .grad themselves (assign when it is None, add otherwise -- what AccumulateGrad does).  A step of the model has ~210
// conv parameters: as many AccumulateGrad nodes less for the autograd engine, whose per-node cost (queueing, stream guards and,
// on this ROCm build, one hipGetDeviceCount call per node) is ~5 us of host time each on a busy box.  Off whenever gradient hooks
// matter (GradReducer's overlap for N > 1 ranks hangs post-accumulate hooks on those nodes).
bool g_direct_grads = false;
void set_direct_grads(bool on) { g_direct_grads = on; }
bool get_direct_grads() { return g_direct_grads; }

void give_grad(const at::Tensor& param, const at::Tensor& gr) {
    if (!gr.defined() || !param.requires_grad()) return;
    at::NoGradGuard ng;
    at::Tensor& cur = const_cast<at::Tensor&>(param).mutable_grad();
    if (!cur.defined()) cur = gr;
    else cur.add_(gr);
}

bool any_requires_grad(at::TensorList xs) {
    for (const at::Tensor& x : xs) if (x.requires_grad()) return true;
    return false;
}

void fill_desc(passion_conv_desc_t& d, const std::vector<int64_t>& geo, const std::vector<int64_t>& seg, double slope,
               const std::vector<OptT>& stats, const std::vector<OptT>& masks, at::TensorList xs, const std::vector<at::Tensor>* dxs,
               int64_t Cout) {
    const int n = (int)xs.size();
    TORCH_CHECK(n >= 1 && n <= PASSION_MAX_SEG && (int)seg.size() == n * SEG_INTS && (int)geo.size() == GEO_INTS, "conv descriptor");
    d.x_dtype = xs[0].scalar_type() == at::kDouble ? PASSION_F32 : dtype_code(xs[0].scalar_type());
    d.y_dtype = (int)geo[7];
    d.B = (int)geo[0]; d.D = (int)geo[1]; d.H = (int)geo[2]; d.W = (int)geo[3];
    d.ksize = (int)geo[4]; d.stride = (int)geo[5]; d.pad_mode = (int)geo[6];
    d.nseg = n;
    d.Cout = (int)Cout;
    d.lrelu_slope = (float)slope;
    for (int i = 0; i < n; ++i) {
        passion_seg_t& sg = d.seg[i];
        const at::Tensor& x = xs[i];
        sg.x = static_cast<const char*>(x.data_ptr()) + seg[i * SEG_INTS + 1] * x.element_size();
        sg.stats = stats[i] ? stats[i]->data_ptr<float>() : nullptr;
        sg.mask = masks[i] ? static_cast<const uint8_t*>(masks[i]->data_ptr()) : nullptr;
        sg.mask_col = (int)seg[i * SEG_INTS + 4];
        sg.mask_ld = masks[i] ? (int)masks[i]->size(1) : 0;
        sg.C = (int)seg[i * SEG_INTS + 0];
        sg.bstride = seg[i * SEG_INTS + 2];
        sg.bmod = (int)seg[i * SEG_INTS + 3];
        sg.dx = (dxs && (*dxs)[i].defined()) ? (*dxs)[i].data_ptr() : nullptr;
        sg.dx_add = nullptr;
        sg.dx_add2 = nullptr;
    }
}

c10::List<OptT> to_list(const std::vector<OptT>& v) {
    c10::List<OptT> l;
    for (const auto& t : v) l.push_back(t);
    return l;
}
std::vector<OptT> from_list(const c10::IValue& iv) {
    std::vector<OptT> v;
    for (const c10::IValue& e : iv.toListRef()) v.push_back(e.isNone() ? OptT() : OptT(e.toTensor()));
    return v;
}

struct ConvBlockNode : public torch::autograd::Function<ConvBlockNode> {
    static variable_list forward(AutogradContext* ctx, const at::Tensor& weight, const OptT& bias, const OptT& residual,
                                 at::TensorList xs, ConvCfg cfg) {
        need_gpu(weight);
        for (const at::Tensor& x : xs) need_gpu(x);
        const int64_t Cout = weight.size(0), ks = cfg.geo[4], stride = cfg.geo[5], pad = ks / 2;
        const int64_t B = cfg.geo[0];
        const int64_t Do = (cfg.geo[1] + 2 * pad - ks) / stride + 1, Ho = (cfg.geo[2] + 2 * pad - ks) / stride + 1,
                      Wo = (cfg.geo[3] + 2 * pad - ks) / stride + 1;
        const at::ScalarType odt = cfg.geo[7] == PASSION_BF16 ? at::kBFloat16 : at::kFloat;
        at::Tensor y = cfg.out ? *cfg.out : at::empty({B, Do, Ho, Wo, Cout}, weight.options().dtype(odt));
        TORCH_CHECK(y.sizes() == at::IntArrayRef({B, Do, Ho, Wo, Cout}) && y.scalar_type() == odt && y.is_contiguous(),
                    "conv_block: `out` has the wrong shape / dtype / layout");
        if (residual)
            TORCH_CHECK(residual->sizes() == y.sizes() && residual->scalar_type() == odt && residual->is_contiguous(), "conv_block: residual");
        if (bias) TORCH_CHECK(bias->scalar_type() == at::kFloat && bias->is_contiguous(), "conv_block: bias must be f32");
        passion_conv_desc_t d;
        fill_desc(d, cfg.geo, cfg.seg, cfg.slope, cfg.stats, cfg.masks, xs, nullptr, Cout);
        at::Tensor st;
        if (cfg.want_stats) {
            st = cfg.stats_out ? *cfg.stats_out : at::empty({B, Cout, 2}, weight.options().dtype(at::kFloat));
            TORCH_CHECK(st.sizes() == at::IntArrayRef({B, Cout, 2}) && st.is_contiguous() && st.scalar_type() == at::kFloat, "conv_block: stats_out");
            check(g.conv_fwd_stats(&d, cfg.packed.data_ptr<float>(), bias ? bias->data_ptr<float>() : nullptr,
                                   residual ? residual->data_ptr() : nullptr, y.data_ptr(), st.data_ptr<float>(), (float)cfg.eps,
                                   cur_stream()),
                  "passion_conv3d_fwd_stats");
        } else {
            check(g.conv_fwd(&d, cfg.packed.data_ptr<float>(), bias ? bias->data_ptr<float>() : nullptr,
                             residual ? residual->data_ptr() : nullptr, y.data_ptr(), cur_stream()),
                  "passion_conv3d_fwd");
        }
        ctx->save_for_backward(xs.vec());
        ctx->saved_data["geo"] = cfg.geo;
        ctx->saved_data["seg"] = cfg.seg;
        ctx->saved_data["slope"] = cfg.slope;
        ctx->saved_data["stats"] = to_list(cfg.stats);
        ctx->saved_data["masks"] = to_list(cfg.masks);
        ctx->saved_data["packed"] = cfg.packed;
        ctx->saved_data["wshape"] = weight.sizes().vec();
        ctx->saved_data["flags"] = std::vector<int64_t>{bias.has_value(), residual.has_value()};
        ctx->saved_data["params"] = c10::List<at::Tensor>(cfg.params);
        if (!cfg.want_stats) return {y};
        ctx->mark_non_differentiable({st});
        ctx->set_materialize_grads(false);       // no zeros_like(stats) (a fill launch) per backward call for the gradient nobody has
        return {y, st};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const variable_list xs = ctx->get_saved_variables();
        const int n = (int)xs.size();
        const std::vector<int64_t> geo = ctx->saved_data["geo"].toIntVector(), seg = ctx->saved_data["seg"].toIntVector(),
                                   wshape = ctx->saved_data["wshape"].toIntVector(), flags = ctx->saved_data["flags"].toIntVector();
        const std::vector<OptT> stats = from_list(ctx->saved_data["stats"]), masks = from_list(ctx->saved_data["masks"]);
        const at::Tensor packed = ctx->saved_data["packed"].toTensor();
        const bool has_bias = flags[0] != 0, has_res = flags[1] != 0;
        // autograd edges exist for the tensor arguments only: weight, [bias], [residual], xs...
        const std::vector<at::Tensor> params = ctx->saved_data["params"].toTensorVector();
        const bool direct = !params.empty();
        size_t e = 0;
        const bool need_w = direct ? params[0].requires_grad() : ctx->needs_input_grad(e);
        ++e;
        const bool need_b = has_bias && (direct ? params[1].requires_grad() : ctx->needs_input_grad(e));
        if (has_bias) ++e;
        const bool need_res = has_res && ctx->needs_input_grad(e);
        if (has_res) ++e;
        if (!grads[0].defined()) return variable_list(4 + n);      // y was not used (set_materialize_grads(false))
        const at::Tensor dy = grads[0].contiguous();
        const auto f32 = dy.options().dtype(at::kFloat);
        at::Tensor dW, db;
        int64_t nW = 1;
        for (int64_t s : wshape) nW *= s;
        if (need_w && need_b) {      // one buffer [dW | db]: the library folds both partial slabs in one launch
            at::Tensor buf = at::empty({nW + wshape[0]}, f32);
            dW = buf.narrow(0, 0, nW).view(wshape);
            db = buf.narrow(0, nW, wshape[0]);
        } else if (need_w) {
            dW = at::empty(wshape, f32);
        } else if (need_b) {
            db = at::empty({wshape[0]}, f32);
        }
        std::vector<at::Tensor> dxs(n);
        // segments that are adjacent slices of one stacked tensor (the four modalities of a grouped encoder level feeding a fusion
        // conv, ops.UnstackFn) get their gradients as adjacent slices of ONE buffer: the unstack node hands that on without a copy
        bool joint = n >= 2;
        for (int i = 0; i < n && joint; ++i) {
            joint = ctx->needs_input_grad(e + i) && xs[i].is_contiguous() && xs[i].sizes() == xs[0].sizes() &&
                    xs[i].scalar_type() == xs[0].scalar_type() && seg[i * SEG_INTS + 1] == 0 &&
                    (i == 0 || static_cast<const char*>(xs[i].data_ptr()) == static_cast<const char*>(xs[i - 1].data_ptr()) + xs[i - 1].nbytes());
        }
        if (joint) {
            std::vector<int64_t> shp = xs[0].sizes().vec();
            const int64_t b0 = shp[0];
            shp[0] = b0 * n;
            at::Tensor all = at::empty(shp, xs[0].options());
            for (int i = 0; i < n; ++i) dxs[i] = all.narrow(0, i * b0, b0);
        } else
        for (int i = 0; i < n; ++i) {
            if (ctx->needs_input_grad(e + i)) {
                TORCH_CHECK(seg[i * SEG_INTS + 1] == 0, "gradient wrt an offset view is not supported");
                dxs[i] = at::empty_like(xs[i]);      // bmod segments: the library sums the passes into the physical samples
            }
        }
        passion_conv_desc_t d;
        fill_desc(d, geo, seg, ctx->saved_data["slope"].toDouble(), stats, masks, xs, &dxs, wshape[0]);
        at::Tensor scratch = at::empty({g.conv_bwd_scratch(&d)}, f32);
        check(g.conv_bwd(&d, packed.data_ptr<float>(), dy.data_ptr(), dW.defined() ? dW.data_ptr<float>() : nullptr,
                         db.defined() ? db.data_ptr<float>() : nullptr, scratch.data_ptr<float>(), cur_stream()),
              "passion_conv3d_bwd");
        variable_list out;
        out.reserve(4 + n);
        if (direct) {
            give_grad(params[0], dW);
            if (has_bias) give_grad(params[1], db);
            dW = at::Tensor(); db = at::Tensor();
        }
        out.push_back(dW);
        out.push_back(db);
        out.push_back(need_res ? dy : at::Tensor());
        for (int i = 0; i < n; ++i) out.push_back(dxs[i]);
        out.push_back(at::Tensor());     // cfg
        return out;
    }
};

// -> [y] or [y, stats of y]
std::vector<at::Tensor> conv_block(const at::Tensor& weight, const OptT& bias, const OptT& residual, const std::vector<at::Tensor>& xs,
                                   std::vector<OptT> stats, std::vector<OptT> masks, std::vector<int64_t> seg, std::vector<int64_t> geo,
                                   double slope, const at::Tensor& packed, const OptT& out, bool want_stats, const OptT& stats_out,
                                   double eps) {
    TORCH_CHECK(g.conv_fwd && g.conv_fwd_stats, "passion_amd host extension is not bound to libpassion_hip.so");
    ConvCfg cfg{std::move(geo), std::move(seg), slope, std::move(stats), std::move(masks), packed, out, want_stats, stats_out, eps, {}};
    // direct-gradient mode needs some OTHER differentiable input to keep the node in the graph (the first conv of a network has none)
    // ... and LEAF parameters: a padded copy of a weight (blocks.run_conv_auto) sends its gradient on through autograd
    if (g_direct_grads && at::GradMode::is_enabled() && (weight.requires_grad() || (bias && bias->requires_grad())) && weight.is_leaf() &&
        (!bias || bias->is_leaf()) && any_requires_grad(xs)) {
        cfg.params.push_back(weight);
        if (bias) cfg.params.push_back(*bias);
        return ConvBlockNode::apply(weight.detach(), bias ? OptT(bias->detach()) : OptT(), residual, at::TensorList(xs), std::move(cfg));
    }
    return ConvBlockNode::apply(weight, bias, residual, at::TensorList(xs), std::move(cfg));
}


// The same block for G weight sets over one stacked batch (passion_conv3d_fwd_grouped: the four modality encoders as one launch
// per layer).  weights / biases: G tensors each (biases may be empty); cfg.packed: the G packed workspaces back to back in ONE
// tensor (equal spacing = one launch).
struct GroupedConvNode : public torch::autograd::Function<GroupedConvNode> {
    static variable_list forward(AutogradContext* ctx, at::TensorList weights, at::TensorList biases, const OptT& residual,
                                 at::TensorList xs, ConvCfg cfg) {
        const int G = (int)weights.size();
        TORCH_CHECK(G >= 1 && G <= 4 && (biases.empty() || (int)biases.size() == G), "grouped conv: 1..4 weight sets");
        for (const at::Tensor& w : weights) { need_gpu(w); TORCH_CHECK(w.sizes() == weights[0].sizes(), "grouped conv: weight shapes differ"); }
        for (const at::Tensor& x : xs) need_gpu(x);
        const at::Tensor& weight = weights[0];
        const int64_t Cout = weight.size(0), ks = cfg.geo[4], stride = cfg.geo[5], pad = ks / 2;
        const int64_t B = cfg.geo[0];
        const int64_t Do = (cfg.geo[1] + 2 * pad - ks) / stride + 1, Ho = (cfg.geo[2] + 2 * pad - ks) / stride + 1,
                      Wo = (cfg.geo[3] + 2 * pad - ks) / stride + 1;
        const at::ScalarType odt = cfg.geo[7] == PASSION_BF16 ? at::kBFloat16 : at::kFloat;
        at::Tensor y = cfg.out ? *cfg.out : at::empty({B, Do, Ho, Wo, Cout}, weight.options().dtype(odt));
        TORCH_CHECK(y.sizes() == at::IntArrayRef({B, Do, Ho, Wo, Cout}) && y.scalar_type() == odt && y.is_contiguous(),
                    "conv_block: `out` has the wrong shape / dtype / layout");
        if (residual)
            TORCH_CHECK(residual->sizes() == y.sizes() && residual->scalar_type() == odt && residual->is_contiguous(), "conv_block: residual");
        TORCH_CHECK(cfg.packed.numel() % G == 0, "grouped conv: packed workspace");
        const int64_t pstride = cfg.packed.numel() / G;
        const float* packed[4];
        const float* bias[4];
        for (int g = 0; g < G; ++g) {
            packed[g] = cfg.packed.data_ptr<float>() + g * pstride;
            if (!biases.empty()) {
                TORCH_CHECK(biases[g].scalar_type() == at::kFloat && biases[g].is_contiguous(), "conv_block: bias must be f32");
                bias[g] = biases[g].data_ptr<float>();
            }
        }
        passion_conv_desc_t d;
        fill_desc(d, cfg.geo, cfg.seg, cfg.slope, cfg.stats, cfg.masks, xs, nullptr, Cout);
        at::Tensor st;
        if (cfg.want_stats) {
            st = cfg.stats_out ? *cfg.stats_out : at::empty({B, Cout, 2}, weight.options().dtype(at::kFloat));
            TORCH_CHECK(st.sizes() == at::IntArrayRef({B, Cout, 2}) && st.is_contiguous() && st.scalar_type() == at::kFloat, "conv_block: stats_out");
        }
        check(g.conv_fwd_grouped(&d, G, packed, biases.empty() ? nullptr : bias, residual ? residual->data_ptr() : nullptr, y.data_ptr(),
                                 cfg.want_stats ? st.data_ptr<float>() : nullptr, (float)cfg.eps, cur_stream()),
              "passion_conv3d_fwd_grouped");
        ctx->save_for_backward(xs.vec());
        ctx->saved_data["geo"] = cfg.geo;
        ctx->saved_data["seg"] = cfg.seg;
        ctx->saved_data["slope"] = cfg.slope;
        ctx->saved_data["stats"] = to_list(cfg.stats);
        ctx->saved_data["masks"] = to_list(cfg.masks);
        ctx->saved_data["packed"] = cfg.packed;
        ctx->saved_data["wshape"] = weight.sizes().vec();
        ctx->saved_data["flags"] = std::vector<int64_t>{G, biases.empty() ? 0 : 1, residual.has_value(), cfg.pass_x};
        ctx->saved_data["params"] = c10::List<at::Tensor>(cfg.params);
        ctx->set_materialize_grads(false);
        // pass_x: xs[0] leaves the node again as its LAST output (autograd makes it a view with this node as grad_fn).  Whoever uses x
        // beside this conv -- the identity branch of x + c3(c2(x)): c3's `residual` -- takes that output, so both gradients of x arrive
        // in ONE backward call and the library adds the second into the write of dx (passion_seg_t.dx_add) instead of autograd
        // summing two tensors in a launch of its own.
        variable_list outs{y};
        if (cfg.want_stats) { ctx->mark_non_differentiable({st}); outs.push_back(st); }
        if (cfg.pass_x >= 1) outs.push_back(xs[0]);
        if (cfg.pass_x >= 2) outs.push_back(xs[0].alias());      // a second reader on another branch (passion_seg_t.dx_add2)
        return outs;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const variable_list xs = ctx->get_saved_variables();
        const int n = (int)xs.size();
        const std::vector<int64_t> geo = ctx->saved_data["geo"].toIntVector(), seg = ctx->saved_data["seg"].toIntVector(),
                                   wshape = ctx->saved_data["wshape"].toIntVector(), flags = ctx->saved_data["flags"].toIntVector();
        const int G = (int)flags[0];
        const bool has_bias = flags[1] != 0, has_res = flags[2] != 0;
        const int pass_x = flags.size() > 3 ? (int)flags[3] : 0;
        const int nout = G + (has_bias ? G : 0) + 1 + n + 1;      // one per forward argument: weights, biases, residual (even absent), xs, cfg
        // the gradients that came in through the handed-on x (the last pass_x outputs)
        at::Tensor xadd, xadd2;
        for (int i = 0; i < pass_x; ++i) {
            const at::Tensor& gr = grads[grads.size() - pass_x + i];
            if (!gr.defined()) continue;
            if (!xadd.defined()) xadd = gr; else xadd2 = gr;
        }
        if (!grads[0].defined()) {
            variable_list none(nout);
            if (xadd.defined()) none[G + (has_bias ? G : 0) + 1] = xadd2.defined() ? xadd + xadd2 : xadd;      // y unused: the identity alone
            return none;
        }
        const std::vector<OptT> stats = from_list(ctx->saved_data["stats"]), masks = from_list(ctx->saved_data["masks"]);
        const at::Tensor packed = ctx->saved_data["packed"].toTensor();
        const std::vector<at::Tensor> params = ctx->saved_data["params"].toTensorVector();
        const bool direct = !params.empty();
        size_t e = 0;
        bool need_w = false;
        for (int g2 = 0; g2 < G; ++g2) need_w = need_w || (direct ? params[g2].requires_grad() : ctx->needs_input_grad(e + g2));
        e += G;
        if (has_bias) { for (int g2 = 0; g2 < G; ++g2) need_w = need_w || (direct ? params[G + g2].requires_grad() : ctx->needs_input_grad(e + g2)); e += G; }
        const bool need_res = has_res && ctx->needs_input_grad(e);
        if (has_res) ++e;
        const at::Tensor dy = grads[0].contiguous();
        const auto f32 = dy.options().dtype(at::kFloat);
        int64_t nW = 1;
        for (int64_t s : wshape) nW *= s;
        const int64_t Cout = wshape[0];
        at::Tensor wbuf;
        float* dWp[4] = {nullptr, nullptr, nullptr, nullptr};
        if (need_w) {
            wbuf = at::empty({G, nW + Cout}, f32);          // [dW | dbias] per weight set
            for (int g2 = 0; g2 < G; ++g2) dWp[g2] = wbuf.data_ptr<float>() + g2 * (nW + Cout);
        }
        std::vector<at::Tensor> dxs(n);
        for (int i = 0; i < n; ++i) {
            if (ctx->needs_input_grad(e + i)) {
                TORCH_CHECK(seg[i * SEG_INTS + 1] == 0, "gradient wrt an offset view is not supported");
                dxs[i] = at::empty_like(xs[i]);
            }
        }
        passion_conv_desc_t d;
        fill_desc(d, geo, seg, ctx->saved_data["slope"].toDouble(), stats, masks, xs, &dxs, Cout);
        bool add_fused = false;
        if (xadd.defined() && dxs[0].defined() && seg[3] == 0 && xadd.sizes() == dxs[0].sizes() && xadd.scalar_type() == dxs[0].scalar_type()) {
            xadd = xadd.contiguous();
            d.seg[0].dx_add = xadd.data_ptr();
            if (xadd2.defined() && xadd2.sizes() == xadd.sizes() && xadd2.scalar_type() == xadd.scalar_type()) {
                xadd2 = xadd2.contiguous();
                d.seg[0].dx_add2 = xadd2.data_ptr();
            } else if (xadd2.defined()) {
                xadd = xadd + xadd2;                     // (cannot happen with tensors handed on by this node: same shape and dtype)
                d.seg[0].dx_add = xadd.data_ptr();
            }
            add_fused = true;
        } else if (xadd.defined() && xadd2.defined()) {
            xadd = xadd + xadd2;
        }
        at::Tensor scratch = at::empty({g.conv_bwd_scratch(&d)}, f32);
        const int64_t pstride = packed.numel() / G;
        const float* pk[4];
        for (int g2 = 0; g2 < G; ++g2) pk[g2] = packed.data_ptr<float>() + g2 * pstride;
        check(g.conv_bwd_grouped(&d, G, pk, dy.data_ptr(), need_w ? dWp : nullptr, scratch.data_ptr<float>(), cur_stream()),
              "passion_conv3d_bwd_grouped");
        variable_list out;
        out.reserve(nout);
        if (direct && need_w) {
            for (int g2 = 0; g2 < G; ++g2) {
                give_grad(params[g2], wbuf[g2].narrow(0, 0, nW).view(wshape));
                if (has_bias) give_grad(params[G + g2], wbuf[g2].narrow(0, nW, Cout));
            }
        }
        for (int g2 = 0; g2 < G; ++g2) out.push_back((need_w && !direct) ? wbuf[g2].narrow(0, 0, nW).view(wshape) : at::Tensor());
        if (has_bias)
            for (int g2 = 0; g2 < G; ++g2) out.push_back((need_w && !direct) ? wbuf[g2].narrow(0, nW, Cout) : at::Tensor());
        out.push_back(need_res ? dy : at::Tensor());
        if (xadd.defined() && !add_fused) dxs[0] = dxs[0].defined() ? dxs[0] + xadd : xadd;      // (shapes the library does not take: autograd's way)
        for (int i = 0; i < n; ++i) out.push_back(dxs[i]);
        out.push_back(at::Tensor());     // cfg
        return out;
    }
};

std::vector<at::Tensor> conv_block_grouped(const std::vector<at::Tensor>& weights, const std::vector<at::Tensor>& biases, const OptT& residual,
                                           const std::vector<at::Tensor>& xs, std::vector<OptT> stats, std::vector<OptT> masks,
                                           std::vector<int64_t> seg, std::vector<int64_t> geo, double slope, const at::Tensor& packed,
                                           const OptT& out, bool want_stats, const OptT& stats_out, double eps, int64_t pass_x) {
    TORCH_CHECK(g.conv_fwd_grouped && g.conv_bwd_grouped, "passion_amd host extension is not bound to libpassion_hip.so");
    ConvCfg cfg{std::move(geo), std::move(seg), slope, std::move(stats), std::move(masks), packed, out, want_stats, stats_out, eps, {}};
    TORCH_CHECK(pass_x >= 0 && pass_x <= 2, "conv_block_grouped: pass_x is 0, 1 or 2");
    cfg.pass_x = (at::GradMode::is_enabled() && any_requires_grad(xs)) ? (int)pass_x : 0;
    const bool handed = cfg.pass_x > 0;
    bool wreq = false, leaves = true;
    for (const at::Tensor& w : weights) { wreq = wreq || w.requires_grad(); leaves = leaves && w.is_leaf(); }
    for (const at::Tensor& b : biases) { wreq = wreq || b.requires_grad(); leaves = leaves && b.is_leaf(); }
    if (g_direct_grads && at::GradMode::is_enabled() && wreq && leaves && any_requires_grad(xs)) {
        std::vector<at::Tensor> wd, bd;
        for (const at::Tensor& w : weights) { cfg.params.push_back(w); wd.push_back(w.detach()); }
        for (const at::Tensor& b : biases) { cfg.params.push_back(b); bd.push_back(b.detach()); }
        std::vector<at::Tensor> r = GroupedConvNode::apply(at::TensorList(wd), at::TensorList(bd), residual, at::TensorList(xs), std::move(cfg));
        for (int i = 0; i < pass_x && !handed; ++i) r.push_back(xs[0]);
        return r;
    }
    std::vector<at::Tensor> r = GroupedConvNode::apply(at::TensorList(weights), at::TensorList(biases), residual, at::TensorList(xs), std::move(cfg));
    for (int i = 0; i < pass_x && !handed; ++i) r.push_back(xs[0]);      // (no graph: x itself)
    return r;
}

at::Tensor instnorm_stats(const at::Tensor& x, double eps, const OptT& out) {
    TORCH_CHECK(g.stats, "passion_amd host extension is not bound to libpassion_hip.so");
    need_gpu(x);
    TORCH_CHECK(x.dim() == 5 && x.is_contiguous(), "instnorm_stats: channels-last (B,D,H,W,C) contiguous tensor expected");
    const int64_t B = x.size(0), V = x.size(1) * x.size(2) * x.size(3), C = x.size(4);
    at::Tensor scratch = at::empty({g.stats_scratch((int)B, V, (int)C)}, x.options().dtype(at::kDouble));
    at::Tensor stats = out ? *out : at::empty({B, C, 2}, x.options().dtype(at::kFloat));
    TORCH_CHECK(stats.sizes() == at::IntArrayRef({B, C, 2}) && stats.is_contiguous() && stats.scalar_type() == at::kFloat, "instnorm_stats: out");
    check(g.stats(x.data_ptr(), dtype_code(x.scalar_type()), (int)B, V, (int)C, V * C, (float)eps, scratch.data_ptr<double>(),
                  stats.data_ptr<float>(), cur_stream()),
          "passion_instnorm_stats");
    return stats;
}

// Optimizer bookkeeping over a few hundred parameters per step, out of the interpreter (the step is host-bound on a busy box):
// (data_ptr of p, data_ptr of p.grad or 0) for every parameter; gradients are checked (f32, GPU) and made contiguous.
std::pair<std::vector<int64_t>, std::vector<int64_t>> param_grad_ptrs(const std::vector<at::Tensor>& params) {
    std::vector<int64_t> pp(params.size()), gp(params.size(), 0);
    for (size_t i = 0; i < params.size(); ++i) {
        const at::Tensor& p = params[i];
        pp[i] = reinterpret_cast<int64_t>(p.data_ptr());
        at::Tensor& g = const_cast<at::Tensor&>(p).mutable_grad();
        if (!g.defined()) continue;
        TORCH_CHECK(g.is_cuda() && g.scalar_type() == at::kFloat, "FusedAdamW needs f32 GPU gradients");
        if (!g.is_contiguous()) g = g.contiguous();
        gp[i] = reinterpret_cast<int64_t>(g.data_ptr());
    }
    return {pp, gp};
}
void drop_grads(const std::vector<at::Tensor>& params) {
    for (const at::Tensor& p : params) const_cast<at::Tensor&>(p).mutable_grad().reset();
}

void bind(uint64_t conv_fwd, uint64_t conv_bwd, uint64_t conv_bwd_scratch, uint64_t stats, uint64_t stats_scratch, uint64_t conv_fwd_stats,
          uint64_t conv_fwd_grouped, uint64_t conv_bwd_grouped) {
    g.conv_fwd_grouped = reinterpret_cast<decltype(g.conv_fwd_grouped)>(conv_fwd_grouped);
    g.conv_bwd_grouped = reinterpret_cast<decltype(g.conv_bwd_grouped)>(conv_bwd_grouped);
    g.conv_fwd = reinterpret_cast<decltype(g.conv_fwd)>(conv_fwd);
    g.conv_fwd_stats = reinterpret_cast<decltype(g.conv_fwd_stats)>(conv_fwd_stats);
    g.conv_bwd = reinterpret_cast<decltype(g.conv_bwd)>(conv_bwd);
    g.conv_bwd_scratch = reinterpret_cast<decltype(g.conv_bwd_scratch)>(conv_bwd_scratch);
    g.stats = reinterpret_cast<decltype(g.stats)>(stats);
    g.stats_scratch = reinterpret_cast<decltype(g.stats_scratch)>(stats_scratch);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("bind", &bind, "addresses of the C-ABI entry points (from the ctypes handle of libpassion_hip.so)");
    m.def("abi_version", []() { return PASSION_ABI_VERSION; }, "the descriptor layout this extension was compiled against (include/passion_hip.h)");
    m.def("conv_block", &conv_block);
    m.def("conv_block_grouped", &conv_block_grouped);
    m.def("param_grad_ptrs", &param_grad_ptrs);
    m.def("drop_grads", &drop_grads);
    m.def("set_direct_grads", &set_direct_grads, "conv nodes write the parameters' .grad themselves (no AccumulateGrad nodes); single-process training only");
    m.def("get_direct_grads", &get_direct_grads);
    m.def("instnorm_stats", &instnorm_stats);
}
