// api.hip -- fused SGD, op-list executor and misc C-ABI entry points of libyunet_hip.so.
#include <string.h>

#include "common.h"

namespace {

// torch.optim.SGD (momentum, dampening, nesterov, weight_decay) over one flat buffer
// (configs/yunet_n.py:1; weight decay hits every parameter, BN gamma/beta included).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, long long n,
                                                  const float* __restrict__ lr_dev, float momentum, float undamped,
                                                  int nesterov, float wd, float gscale, int first) {
    const float lr = lr_dev[0];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // one element per thread: no loop (test_isa_guard)
    if (i >= n) return;
    const float w = p[i];
    const float d = g[i] * gscale + wd * w;
    float step = d;
    if (momentum != 0.0f) {
        const float b = first ? d : buf[i] * momentum + undamped * d;
        buf[i] = b;
        step = nesterov ? d + momentum * b : b;
    }
    p[i] = w - lr * step;
}

// every BatchNorm layer of the model in one launch (one workgroup per layer)
__global__ void bn_batch_kernel(const int32_t* __restrict__ table, const double* __restrict__ stats_base,
                                float* __restrict__ rm, float* __restrict__ rv, float momentum,
                                float* __restrict__ grad_base, int mode) {
    const int32_t* t = table + blockIdx.x * 7;
    const int C = t[1], count = t[2], slots = t[6];
    const double* st = stats_base + t[0];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (mode == 0) {
            const double inv = 1.0 / (double)count;
            const double mean = bn_sum(st, slots, C, c) * inv;
            double var = bn_sum(st, slots, C, C + c) * inv - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double unb = count > 1 ? var * ((double)count / (double)(count - 1)) : var;
            float* m = rm + t[3];
            float* v = rv + t[3];
            m[c] = (1.0f - momentum) * m[c] + momentum * (float)mean;
            v[c] = (1.0f - momentum) * v[c] + momentum * (float)unb;
        } else if (mode == 1) {
            grad_base[t[5] + c] = (float)bn_sum(st, slots, C, c);        // d(beta)  = sum dy
            grad_base[t[4] + c] = (float)bn_sum(st, slots, C, C + c);    // d(gamma) = sum dy * xhat
        } else {
            // mode 2, eval(): synthesise the sums whose mean / biased variance are the RUNNING
            // statistics, so the train-mode kernels (which derive their coefficients from the
            // producer's sums) apply nn.BatchNorm2d in eval mode unchanged
            double* w = const_cast<double*>(st);
            const double m = (double)rm[t[3] + c], v = (double)rv[t[3] + c];
            w[c] = m * (double)count;
            w[C + c] = (v + m * m) * (double)count;
            for (int k = 1; k < slots; ++k) w[k * 2 * C + c] = 0.0, w[k * 2 * C + C + c] = 0.0;
        }
    }
}

}  // namespace

extern "C" int yunet_bn_batch(const int32_t* table, int n, const double* stats_base,
                              float* running_mean, float* running_var, float momentum,
                              float* grad_base, int mode, void* stream) {
    if (n < 1) return YUNET_EINVAL;
    hipLaunchKernelGGL(bn_batch_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, table, stats_base,
                       running_mean, running_var, momentum, grad_base, mode);
    return hip_status();
}

// one float4 per thread, no loop (a grid-stride loop here is compiled as one memory round trip per trip: tests/test_isa_guard.py)
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, size_t n, int vec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i < n / 4) {
            const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
            reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        } else if (i - n / 4 < n % 4) {
            const size_t k = (n / 4) * 4 + (i - n / 4);
            out[k] = a[k] + b[k];
        }
    } else if (i < n) {
        out[i] = a[i] + b[i];
    }
}

extern "C" int yunet_add(const float* a, const float* b, float* out, size_t n, void* stream) {
    if (!a || !b || !out) return YUNET_EINVAL;
    if (n == 0) return 0;
    const int vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const size_t threads = vec ? n / 4 + n % 4 : n;
    const size_t blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffull) return YUNET_EINVAL;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, n, vec);
    return hip_status();
}

extern "C" int yunet_sgd_step_ex(float* params, const float* grads, float* momentum_buf, int64_t n,
                                 const float* lr_dev, float momentum, float dampening, int nesterov,
                                 float weight_decay, float grad_scale, int first_step, void* stream) {
    if (n < 1 || !params || !grads || !lr_dev) return YUNET_EINVAL;
    if (momentum != 0.0f && !momentum_buf) return YUNET_EINVAL;
    if (nesterov && (momentum <= 0.0f || dampening != 0.0f)) return YUNET_EINVAL;      // torch: ValueError
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffll) return YUNET_EINVAL;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads,
                       momentum_buf, (long long)n, lr_dev, momentum, 1.0f - dampening, nesterov ? 1 : 0, weight_decay,
                       grad_scale, first_step);
    return hip_status();
}

extern "C" int yunet_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n,
                              const float* lr_dev, float momentum, float weight_decay,
                              float grad_scale, int first_step, void* stream) {
    return yunet_sgd_step_ex(params, grads, momentum_buf, n, lr_dev, momentum, 0.0f, 0, weight_decay, grad_scale,
                             first_step, stream);
}

// ---- measurement switches (common.h: YunetOptions) --------------------------------------------------------------
YunetOptions& yunet_options() {
    static YunetOptions o = [] {
        auto env = [](const char* k, int dflt) {
            const char* e = getenv(k);
            return e ? atoi(e) : dflt;
        };
        YunetOptions v;
        v.no_pack = getenv("YUNET_NO_PACK") ? 1 : 0;
        v.bwd_fp32mma = getenv("YUNET_BWD_FP32MMA") ? 1 : 0;
        v.bwd64_nw = env("YUNET_BWD64_NW", 0);
        v.ew_grid = env("YUNET_EW_GRID", 768);
        v.fwd_blocks_per_cu = env("YUNET_DP_FWD_BLOCKS_PER_CU", 0);
        v.fwd64s = env("YUNET_FWD64S", 2);
        v.fwd64s_rows = env("YUNET_FWD64S_ROWS", 0);
        v.bwd16s = env("YUNET_BWD16S", 1);
        v.bwd16s_rows = env("YUNET_BWD16S_ROWS", 0);
        v.stem_mma = env("YUNET_STEM_MMA", 1);
        v.fwd16s = env("YUNET_FWD16S", 1);
        v.upadd_coarse = env("YUNET_UPADD_COARSE", 1);
        v.bwd32_split = env("YUNET_BWD32_SPLIT", 1);
        v.assign_v2 = env("YUNET_ASSIGN_V2", 1);
        v.fwd_group = env("YUNET_FWD_GROUP", 1);
        v.oneshot_timeout_ms = env("YUNET_ONESHOT_TIMEOUT_MS", 120000);
        if (v.bwd64_nw != 4 && v.bwd64_nw != 8) v.bwd64_nw = 0;
        if (v.ew_grid < 1) v.ew_grid = 768;
        return v;
    }();
    return o;
}
int yunet_option_assign_v2() { return yunet_options().assign_v2; }
extern "C" int yunet_set_option(const char* name, int value) {
    YunetOptions& o = yunet_options();
    int* slot = nullptr;
    if (!name) return YUNET_EINVAL;
    if (!strcmp(name, "no_pack")) slot = &o.no_pack;
    else if (!strcmp(name, "bwd_fp32mma")) slot = &o.bwd_fp32mma;
    else if (!strcmp(name, "bwd64_nw")) slot = &o.bwd64_nw;
    else if (!strcmp(name, "ew_grid")) slot = &o.ew_grid;
    else if (!strcmp(name, "fwd_blocks_per_cu")) slot = &o.fwd_blocks_per_cu;
    else if (!strcmp(name, "fwd64s")) slot = &o.fwd64s;
    else if (!strcmp(name, "fwd64s_rows")) slot = &o.fwd64s_rows;
    else if (!strcmp(name, "bwd16s")) slot = &o.bwd16s;
    else if (!strcmp(name, "bwd16s_rows")) slot = &o.bwd16s_rows;
    else if (!strcmp(name, "stem_mma")) slot = &o.stem_mma;
    else if (!strcmp(name, "fwd16s")) slot = &o.fwd16s;
    else if (!strcmp(name, "upadd_coarse")) slot = &o.upadd_coarse;
    else if (!strcmp(name, "bwd32_split")) slot = &o.bwd32_split;
    else if (!strcmp(name, "assign_v2")) slot = &o.assign_v2;
    else if (!strcmp(name, "fwd_group")) slot = &o.fwd_group;
    else if (!strcmp(name, "oneshot_timeout_ms")) slot = &o.oneshot_timeout_ms;
    if (!slot || value < 0) return YUNET_EINVAL;
    if (slot == &o.bwd64_nw && value != 0 && value != 4 && value != 8) return YUNET_EINVAL;
    if (slot == &o.ew_grid && value == 0) value = 768;
    const int prev = *slot;
    *slot = value;
    return prev;
}

// the same conv entry points compiled with bf16 activation storage (conv_fwd.hip / conv_bwd.hip -DYUNET_ACT_BF16)
extern "C" {
int yunet_stem_fwd_bf16(const float*, const float*, const float*, float*, double*, int, int, int, int, void*);
int yunet_stem_bwd_bf16(const float*, const float*, const float*, const YunetBN*, float*, int, int, int, int, int, void*);
int yunet_dp_fwd_bf16(const YunetDP*, void*);
int yunet_dp_fwd_group_bf16(const YunetDP* const*, int, void*);
int yunet_dp_bwd_bf16(const YunetDP*, void*);
int yunet_pool_fwd_bf16(const float*, const YunetBN*, float*, int, int, int, int, void*);
int yunet_pool_bwd_bf16(const float*, const YunetBN*, const float*, float*, int, int, int, int, int, void*);
int yunet_upadd_fwd_bf16(const float*, const YunetBN*, const float*, const YunetBN*, float*, int, int, int, int, void*);
int yunet_upadd_bwd_bf16(const float*, const YunetBN*, const float*, const YunetBN*, const float*, float*, int, float*,
                         int, int, int, int, int, void*);
}

// Op lists select the activation storage type per op: YunetDP.x_dtype for the ConvDPUnits, i[11] for
// the stem / pool / upsample-add ops (0 = fp32, 1 = bf16).
// ---- lanes: side streams + events of the executor, one set per (host thread, device), created on first use: two
// threads (or two devices) replaying op lists never share an event, and a side stream always belongs to the device
// that is current in the calling thread
static int g_lanes_on = 1;
struct LaneSet {
    bool ready;
    hipStream_t side[YUNET_MAX_LANES];
    hipEvent_t fork_ev, join_ev[YUNET_MAX_LANES];
};
static constexpr int kMaxDevices = 16;
static thread_local LaneSet t_lanes[kMaxDevices];
static LaneSet* lanes_get() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    LaneSet& s = t_lanes[dev];
    if (s.ready) return &s;
    for (int l = 0; l < YUNET_MAX_LANES; ++l) {
        if (hipStreamCreateWithFlags(&s.side[l], hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&s.join_ev[l], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    if (hipEventCreateWithFlags(&s.fork_ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    s.ready = true;
    return &s;
}
extern "C" int yunet_exec_lanes(int enable) {
    const int prev = g_lanes_on;
    g_lanes_on = enable ? 1 : 0;
    return prev;
}

extern "C" int yunet_exec(const YunetOp* ops, int n_ops, void* main_stream) {
    unsigned open_lanes = 0;
    LaneSet* ls = nullptr;
    for (int k = 0; k < n_ops; ++k) {
        const YunetOp& o = ops[k];
        int rc = 0;
        void* stream = main_stream;
        if (o.opcode == YUNET_OP_FORK || o.opcode == YUNET_OP_JOIN) {
            if (!g_lanes_on) continue;
            if (!ls && !(ls = lanes_get())) return YUNET_EINVAL * 1000 - k;
            const unsigned mask = (unsigned)o.i[0] & ((1u << (YUNET_MAX_LANES + 1)) - 2u);
            if (o.opcode == YUNET_OP_FORK) {
                if (hipEventRecord(ls->fork_ev, (hipStream_t)main_stream) != hipSuccess) return YUNET_EINVAL * 1000 - k;
                for (int l = 1; l <= YUNET_MAX_LANES; ++l)
                    if (mask & (1u << l)) hipStreamWaitEvent(ls->side[l - 1], ls->fork_ev, 0);
                open_lanes |= mask;
            } else {
                for (int l = 1; l <= YUNET_MAX_LANES; ++l)
                    if (mask & (1u << l)) {
                        hipEventRecord(ls->join_ev[l - 1], ls->side[l - 1]);
                        hipStreamWaitEvent((hipStream_t)main_stream, ls->join_ev[l - 1], 0);
                    }
                open_lanes &= ~mask;
            }
            continue;
        }
        const int lane = o.i[YUNET_OP_LANE];
        if (g_lanes_on && lane > 0) {
            if (lane > YUNET_MAX_LANES || !(open_lanes & (1u << lane)) || !ls) return YUNET_EINVAL * 1000 - k;   // not forked
            stream = (void*)ls->side[lane - 1];
        }
        switch (o.opcode) {
            case YUNET_OP_STEM_FWD:
                rc = (o.i[11] == YUNET_BF16 ? yunet_stem_fwd_bf16 : yunet_stem_fwd)((const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2],
                                    (float*)o.p[3], (double*)o.p[4], o.i[0], o.i[1], o.i[2], o.i[3],
                                    stream);
                break;
            case YUNET_OP_STEM_BWD:
                // p[4], p[5] = the stem's weights / bias: the weight gradient recomputes z from the image
                if (o.p[4] && o.p[5] && yunet_options().stem_mma) {        // (reads the image and dy only: any activation storage type)
                    rc = yunet_stem_bwd_rz((const float*)o.p[0], (const float*)o.p[4], (const float*)o.p[5], (const float*)o.p[2],
                                           &o.bn[0], (float*)o.p[3], o.i[4], o.i[0], o.i[1], o.i[2], o.i[3], stream);
                    break;
                }
                rc = (o.i[11] == YUNET_BF16 ? yunet_stem_bwd_bf16 : yunet_stem_bwd)((const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2],
                                    &o.bn[0], (float*)o.p[3], o.i[4], o.i[0], o.i[1], o.i[2], o.i[3],
                                    stream);
                break;
            case YUNET_OP_DP_FWD: {
                // a group of independent units (include/yunet_hip.h: YUNET_OP_GROUP): one call, and where the kernels allow
                // it one launch
                const int g = o.i[YUNET_OP_GROUP];
                if (g >= 2 && g <= YUNET_DP_GROUP_MAX && k + g <= n_ops) {
                    const YunetDP* units[YUNET_DP_GROUP_MAX];
                    bool ok = true;
                    for (int j = 0; j < g; ++j) {
                        const YunetOp& oj = ops[k + j];
                        ok = ok && oj.opcode == YUNET_OP_DP_FWD && oj.i[YUNET_OP_LANE] == lane && oj.dp.x_dtype == o.dp.x_dtype;
                        units[j] = &oj.dp;
                    }
                    if (!ok) return YUNET_EINVAL * 1000 - k;
                    rc = o.dp.x_dtype == YUNET_BF16 ? yunet_dp_fwd_group_bf16(units, g, stream) : yunet_dp_fwd_group(units, g, stream);
                    if (rc == 0) k += g - 1;
                    break;
                }
                rc = o.dp.x_dtype == YUNET_BF16 ? yunet_dp_fwd_bf16(&o.dp, stream) : yunet_dp_fwd(&o.dp, stream);
                break;
            }
            case YUNET_OP_DP_BWD:
                rc = o.dp.x_dtype == YUNET_BF16 ? yunet_dp_bwd_bf16(&o.dp, stream) : yunet_dp_bwd(&o.dp, stream);
                break;
            case YUNET_OP_POOL_FWD:
                rc = (o.i[11] == YUNET_BF16 ? yunet_pool_fwd_bf16 : yunet_pool_fwd)((const float*)o.p[0], &o.bn[0], (float*)o.p[1], o.i[0], o.i[1],
                                    o.i[2], o.i[3], stream);
                break;
            case YUNET_OP_POOL_BWD:
                // p[3]: the upsample-add's share of the same gradient (NULL: plain max_pool2d backward)
                rc = (o.i[11] == YUNET_BF16 ? yunet_pool_bwd_add_bf16 : yunet_pool_bwd_add)(
                    (const float*)o.p[0], &o.bn[0], (const float*)o.p[1], (const float*)o.p[3], (float*)o.p[2], o.i[4], o.i[0],
                    o.i[1], o.i[2], o.i[3], stream);
                break;
            case YUNET_OP_UPADD_FWD:
                rc = (o.i[11] == YUNET_BF16 ? yunet_upadd_fwd_bf16 : yunet_upadd_fwd)((const float*)o.p[0], &o.bn[0], (const float*)o.p[1], &o.bn[1],
                                     (float*)o.p[2], o.i[0], o.i[1], o.i[2], o.i[3], stream);
                break;
            case YUNET_OP_UPADD_BWD:
                rc = (o.i[11] == YUNET_BF16 ? yunet_upadd_bwd_bf16 : yunet_upadd_bwd)((const float*)o.p[0], &o.bn[0], (const float*)o.p[1], &o.bn[1],
                                     (const float*)o.p[2], (float*)o.p[3], o.i[4], (float*)o.p[4],
                                     o.i[5], o.i[0], o.i[1], o.i[2], o.i[3], stream);
                break;
            case YUNET_OP_BN_RUNNING:
                rc = yunet_bn_update_running((const double*)o.p[0], (float*)o.p[1], (float*)o.p[2],
                                             o.i[0], o.i[1], o.f[0], stream);
                break;
            case YUNET_OP_BN_PARAM_GRAD:
                rc = yunet_bn_param_grad((const double*)o.p[0], (float*)o.p[1], (float*)o.p[2], o.i[0],
                                         o.i[1], stream);
                break;
            case YUNET_OP_REDUCE_PARTIALS:
                rc = yunet_reduce_partials((const float*)o.p[0], o.i[0], o.i[1], (float*)o.p[1],
                                           o.i[2], stream);
                break;
            case YUNET_OP_REDUCE_BATCH:
                rc = yunet_reduce_partials_batch((const YunetReduceJob*)o.p[0], o.i[0], o.i[1], stream);
                break;
            case YUNET_OP_ASSIGN: {
                const bool dflt = o.f[1] == 0.0f && o.f[2] == 0.0f;
                const YunetAssignCfg ac{o.f[0], o.i[3] ? o.i[3] : 10, dflt ? 3.0f : o.f[1], dflt ? 1.0f : o.f[2]};
                rc = yunet_assign_cfg((const float*)o.p[0], nullptr, nullptr, (const float*)o.p[1], (const float*)o.p[2],
                                      (const int32_t*)o.p[3], (const int32_t*)o.p[4], &o.lv, o.i[0], o.i[1],
                                      o.i[2], &ac, (int32_t*)o.p[5], (int32_t*)o.p[6], (float*)o.p[7],
                                      (float*)o.p[8], (float*)o.p[9], stream);
                break;
            }
            case YUNET_OP_LOSS_NORM:
                rc = yunet_loss_norm((const float*)o.p[0], o.i[0], o.f[0], (float*)o.p[1], stream);
                break;
            case YUNET_OP_LOSS:
                rc = yunet_loss((const float*)o.p[0], (const int32_t*)o.p[1], (const float*)o.p[2],
                                (const float*)o.p[3], (const float*)o.p[4], &o.lv, &o.loss,
                                (const float*)o.p[5], o.i[0], o.i[1], o.i[2], (float*)o.p[6],
                                (float*)o.p[7], o.i[3], stream);
                break;
            case YUNET_OP_LOSS_FINALIZE:
                rc = yunet_loss_finalize_ex((const float*)o.p[0], o.i[0], (float*)o.p[1], (float*)o.p[2],
                                            (const float*)o.p[3], (float*)o.p[4], stream);
                break;
            case YUNET_OP_SGD: {
                const int64_t n = ((int64_t)(uint32_t)o.i[1] << 32) | (uint32_t)o.i[0];
                rc = yunet_sgd_step((float*)o.p[0], (const float*)o.p[1], (float*)o.p[2], n,
                                    (const float*)o.p[3], o.f[0], o.f[1], o.f[2], o.i[2], stream);
                break;
            }
            case YUNET_OP_MEMSET: {
                const size_t n = ((size_t)(uint32_t)o.i[1] << 32) | (uint32_t)o.i[0];
                rc = -(int)hipMemsetAsync(o.p[0], 0, n, (hipStream_t)stream);
                break;
            }
            case YUNET_OP_ADD: {
                const size_t n = ((size_t)(uint32_t)o.i[1] << 32) | (uint32_t)o.i[0];
                rc = yunet_add((const float*)o.p[0], (const float*)o.p[1], (float*)o.p[2], n, stream);
                break;
            }
            case YUNET_OP_BN_BATCH:
                rc = yunet_bn_batch((const int32_t*)o.p[0], o.i[0], (const double*)o.p[1], (float*)o.p[2],
                                    (float*)o.p[3], o.f[0], (float*)o.p[4], o.i[1], stream);
                break;
            default:
                return YUNET_EOPCODE;
        }
        if (rc != 0) return rc < 0 ? rc * 1000 - k : -(rc * 1000 + k);
    }
    if (open_lanes) {      // a list must not end with work pending on a side stream: join defensively, report
        for (int l = 1; l <= YUNET_MAX_LANES; ++l)
            if (open_lanes & (1u << l)) {
                hipEventRecord(ls->join_ev[l - 1], ls->side[l - 1]);
                hipStreamWaitEvent((hipStream_t)main_stream, ls->join_ev[l - 1], 0);
            }
        return YUNET_EINVAL * 1000 - n_ops;
    }
    return 0;
}
