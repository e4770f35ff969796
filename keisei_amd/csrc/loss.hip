// Fused KataGo-PPO minibatch loss: masked log-softmax over the 11 259 spatial actions, log-prob
// gather, clipped surrogate, entropy over legal actions, W/D/L cross-entropy (ignore_index -1,
// all-ignored guard), score MSE -- forward values AND the gradients w.r.t. the three model
// outputs in one pass, with the reference's two guards (NaN logits / zero legal actions) as
// device flags instead of host syncs.
//
// policy_loss_kernel: one workgroup per sample; the 45 KB logit row and its legal mask are
// staged once in LDS, so HBM sees one read of logits+mask and one write of dlogits.  Rows,
// masks and per-sample scalars are fetched through the minibatch index vector, i.e. the
// reference's gather (katago_ppo.py:835-841) is fused into the consumers.
//
// Reference: keisei/training/katago_ppo.py:33-57 (ppo_clip_loss, wdl_cross_entropy_loss),
// :857-924 (loss assembly), value_adapter.py:98-126; gradients follow torch autograd
// (torch.minimum ties split 1/2-1/2, clamp passes gradient on the closed interval).
#include "common.h"

namespace {

constexpr int kPolThreads = 256;

struct PolicyArgs {
    const float* logits;      // (B,A)
    const uint8_t* legal;     // (S,A) bool rows, or -- legal_words > 0 -- (S,legal_words) uint32 rows, bit j of word w =
                              // action 32 w + j (the device rollout store's packing, rollout.hip); row idx[b]
    const long long* actions; // (S)
    const float* old_lp;      // (S)
    const float* adv;         // (S)
    const long long* idx;     // (B) or null (identity)
    float* dlogits;           // (B,A) or null
    float* new_lp;            // (B)
    float* rowloss;           // (B)  -min(surr1,surr2)
    float* rowent;            // (B)
    int* flags;               // [0]=NaN in logits, [1] bit 0 = zero legal actions, bit 1 = action id outside [0, A)
    const float* gscale;      // device scalar: loss scale (GradScaler) or null (=1)
    float clip_eps, w_policy, w_entropy;   // w_* already divided by B
    int A, legal_words;
};

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o); v = is_max ? fmaxf(v, t) : v + t; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < kPolThreads / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

__global__ __launch_bounds__(kPolThreads) void policy_loss_kernel(PolicyArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* row = reinterpret_cast<float*>(smem);
    uint8_t* msk = reinterpret_cast<uint8_t*>(smem + ((size_t)a.A * 4 + 15) / 16 * 16);
    __shared__ float red[kPolThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long src = a.idx ? a.idx[b] : b;
    const float* lg = a.logits + (size_t)b * a.A;
    const uint8_t* lm = a.legal + (size_t)src * a.A;
    const uint32_t* lw = reinterpret_cast<const uint32_t*>(a.legal) + (size_t)src * a.legal_words;

    float mx = -INFINITY, nlegal = 0.f;
    int nan_seen = 0;
    // four logits and mask entries are requested before the first is used (the row is one HBM round trip per pass otherwise)
    for (int j0 = tid; j0 < a.A; j0 += 4 * kPolThreads) {
        float v[4]; uint8_t k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * kPolThreads;
            v[u] = j < a.A ? lg[j] : 0.f;
            k[u] = j < a.A ? (a.legal_words ? (uint8_t)((lw[j >> 5] >> (j & 31)) & 1u) : lm[j]) : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * kPolThreads;
            if (j >= a.A) break;
            row[j] = v[u]; msk[j] = k[u];
            nan_seen |= (v[u] != v[u]);
            if (k[u]) { mx = fmaxf(mx, v[u]); nlegal += 1.f; }
        }
    }
    mx = block_reduce(mx, red, true);
    nlegal = block_reduce(nlegal, red, false);
    const float nanf_ = block_reduce((float)nan_seen, red, false);
    if (tid == 0) {
        if (nanf_ > 0.f) atomicOr(&a.flags[0], 1);
        if (nlegal == 0.f) atomicOr(&a.flags[1], 1);
    }
    float s = 0.f, t = 0.f;
    for (int j = tid; j < a.A; j += kPolThreads)
        if (msk[j]) { const float e = expf(row[j] - mx); s += e; t += e * row[j]; }
    s = block_reduce(s, red, false);
    t = block_reduce(t, red, false);
    const float lse = mx + logf(s);
    const float H = lse - t / s;
    // an action id outside [0, A) (the reference's gather would trap on the device) is flagged and read as action 0
    const long long act_raw = a.actions[src];
    const bool act_ok = act_raw >= 0 && act_raw < a.A;
    if (!act_ok && tid == 0) atomicOr(&a.flags[1], 2);
    const long long act = act_ok ? act_raw : 0;
    const float nlp = msk[act] ? row[act] - lse : -INFINITY;
    const float adv = a.adv[src];
    const float ratio = expf(nlp - a.old_lp[src]);
    const float lo = 1.f - a.clip_eps, hi = 1.f + a.clip_eps;
    const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, lo), hi) * adv;
    const bool inrange = ratio >= lo && ratio <= hi;
    float gr;                                   // d min(s1,s2) / d ratio
    if (s1 < s2) gr = adv;
    else if (s1 > s2) gr = inrange ? adv : 0.f;
    else gr = 0.5f * adv + (inrange ? 0.5f * adv : 0.f);
    if (tid == 0) { a.new_lp[b] = nlp; a.rowloss[b] = -fminf(s1, s2); a.rowent[b] = H; }
    if (a.dlogits) {
        const float gs = a.gscale ? *a.gscale : 1.f;
        const float dnlp = -gr * ratio * a.w_policy * gs;     // dL/d new_log_prob
        const float we = a.w_entropy * gs;
        float* dl = a.dlogits + (size_t)b * a.A;
        for (int j = tid; j < a.A; j += kPolThreads) {
            float g = 0.f;
            if (msk[j]) {
                const float lp = row[j] - lse, p = expf(lp);
                g = -dnlp * p + we * p * (lp + H);
                if (j == act) g += dnlp;
            }
            dl[j] = g;
        }
    }
}

// Rollout action selection (katago_ppo.py:566-584): probs[b] = softmax of the logits over the legal actions (0 elsewhere),
// nlegal[b] = number of legal actions.  One pass over the LDS-staged row replaces the reference's sum / masked_fill /
// softmax / renormalise / clamp / log chain of ~10 launches; sampling itself stays torch.multinomial on these rows.
// legal: bool rows (legal_words == 0) or packed rows as in ka_policy_loss.  flags[0] |= NaN in the logits.
__global__ __launch_bounds__(kPolThreads) void masked_softmax_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ legal,
                                                                     float* __restrict__ probs, int* __restrict__ nlegal_out,
                                                                     int* __restrict__ flags, int A, int legal_words) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* row = reinterpret_cast<float*>(smem);
    uint8_t* msk = reinterpret_cast<uint8_t*>(smem + ((size_t)A * 4 + 15) / 16 * 16);
    __shared__ float red[kPolThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (size_t)b * A;
    const uint8_t* lm = legal + (size_t)b * A;
    const uint32_t* lw = reinterpret_cast<const uint32_t*>(legal) + (size_t)b * legal_words;
    float mx = -INFINITY, nlegal = 0.f;
    int nan_seen = 0;
    for (int j = tid; j < A; j += kPolThreads) {
        const float v = lg[j];
        const uint8_t k = legal_words ? (uint8_t)((lw[j >> 5] >> (j & 31)) & 1u) : lm[j];
        row[j] = v; msk[j] = k;
        nan_seen |= (v != v);
        if (k) { mx = fmaxf(mx, v); nlegal += 1.f; }
    }
    mx = block_reduce(mx, red, true);
    nlegal = block_reduce(nlegal, red, false);
    const float nanf_ = block_reduce((float)nan_seen, red, false);
    float s = 0.f;
    for (int j = tid; j < A; j += kPolThreads)
        if (msk[j]) s += expf(row[j] - mx);
    s = block_reduce(s, red, false);
    if (tid == 0) {
        if (nanf_ > 0.f) atomicOr(&flags[0], 1);
        nlegal_out[b] = (int)nlegal;
    }
    const float inv = 1.f / s;
    float* pr = probs + (size_t)b * A;
    for (int j = tid; j < A; j += kPolThreads) pr[j] = msk[j] ? expf(row[j] - mx) * inv : 0.f;
}

// Supervised policy head (keisei/sl/trainer.py:150-152, F.cross_entropy over all A actions, mean over the batch): one
// workgroup per sample stages the logit row in LDS, rowloss[b] = logsumexp(row) - row[target[b]] and
// dlogits = w * (softmax(row) - onehot(target)), w = lambda_policy / B (times the loss scale).  flags[0] |= NaN in the logits,
// flags[1] |= a target outside [0, A).
struct PolicyCeArgs {
    const float* logits; const long long* targets; const long long* idx; float* dlogits; float* rowloss; int* flags;
    const float* gscale; float w_policy; int A;
};

__global__ __launch_bounds__(kPolThreads) void policy_ce_kernel(PolicyCeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* row = reinterpret_cast<float*>(smem);
    __shared__ float red[kPolThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long src = a.idx ? a.idx[b] : b;
    const float* lg = a.logits + (size_t)b * a.A;
    float mx = -INFINITY;
    int nan_seen = 0;
    for (int j = tid; j < a.A; j += kPolThreads) {
        const float v = lg[j];
        row[j] = v;
        nan_seen |= (v != v);
        mx = fmaxf(mx, v);
    }
    mx = block_reduce(mx, red, true);
    const float nanf_ = block_reduce((float)nan_seen, red, false);
    float s = 0.f;
    for (int j = tid; j < a.A; j += kPolThreads) s += expf(row[j] - mx);
    s = block_reduce(s, red, false);
    const float lse = mx + logf(s);
    const long long tgt = a.targets[src];
    const bool ok = tgt >= 0 && tgt < a.A;
    if (tid == 0) {
        if (nanf_ > 0.f) atomicOr(&a.flags[0], 1);
        if (!ok) atomicOr(&a.flags[1], 1);
        a.rowloss[b] = ok ? lse - row[tgt] : 0.f;
    }
    if (a.dlogits) {
        const float w = a.w_policy * (a.gscale ? *a.gscale : 1.f);
        float* dl = a.dlogits + (size_t)b * a.A;
        for (int j = tid; j < a.A; j += kPolThreads) {
            float g = w * expf(row[j] - lse);
            if (j == tgt) g -= w;
            dl[j] = g;
        }
    }
}

struct ValueArgs {
    const float* vlogits;      // (B,3)
    const float* score;        // (B)
    const long long* cats;     // (S)
    const float* targets;      // (S)
    const long long* idx;      // (B) or null
    const float* rowloss;      // (B) from policy kernel
    const float* rowent;       // (B)
    float* dvlogits;           // (B,3) or null
    float* dscore;             // (B)
    float* out;                // [8]: policy, value_ce, score_mse, entropy, total, value_acc_count.. see below
    float* acc;                // [4] running sums: policy, value(combined or ce), score, entropy   (or null)
    const float* gscale;
    float lambda_policy, lambda_value, lambda_score, entropy_coeff;
    int B, combined_value_metric;
};

// single workgroup: B is a minibatch (<= a few 10^4); fixed-order reductions
__global__ __launch_bounds__(1024) void value_loss_kernel(ValueArgs a) {
    __shared__ double red[8][16];
    const int tid = threadIdx.x;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // nvalid, ce, mse, pl, ent, correct, pred0, pred1
    for (int b = tid; b < a.B; b += 1024) {
        const long long src = a.idx ? a.idx[b] : b;
        const float l0 = a.vlogits[b * 3], l1 = a.vlogits[b * 3 + 1], l2 = a.vlogits[b * 3 + 2];
        const long long cat = a.cats[src];
        const float m = fmaxf(l0, fmaxf(l1, l2));
        const float lse = m + logf(expf(l0 - m) + expf(l1 - m) + expf(l2 - m));
        if (cat >= 0) {
            acc[0] += 1.0;
            acc[1] += (double)(lse - (cat == 0 ? l0 : (cat == 1 ? l1 : l2)));
            const int pred = (l0 >= l1 && l0 >= l2) ? 0 : (l1 >= l2 ? 1 : 2);
            acc[5] += pred == cat; acc[6] += pred == 0; acc[7] += pred == 1;
        }
        const float d = a.score[b] - a.targets[src];
        acc[2] += (double)(d * d);
        acc[3] += (double)a.rowloss[b];
        acc[4] += (double)a.rowent[b];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc[k] = wave_sum_d(acc[k]);
        if ((tid & 63) == 0) red[k][tid >> 6] = acc[k];
    }
    __syncthreads();
    double tot[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { tot[k] = 0; for (int w = 0; w < 16; ++w) tot[k] += red[k][w]; }
    const double nvalid = tot[0];
    const float ce = nvalid > 0 ? (float)(tot[1] / nvalid) : 0.f;
    const float mse = (float)(tot[2] / a.B), pl = (float)(tot[3] / a.B), ent = (float)(tot[4] / a.B);
    if (tid == 0) {
        const float vs = a.lambda_value * ce + a.lambda_score * mse;
        a.out[0] = pl; a.out[1] = ce; a.out[2] = mse; a.out[3] = ent;
        a.out[4] = a.lambda_policy * pl + vs - a.entropy_coeff * ent;
        a.out[5] = (float)nvalid;
        a.out[6] = nvalid > 0 ? (float)(tot[5] / nvalid) : 0.f;            // value_accuracy
        a.out[7] = nvalid > 0 ? (float)(tot[6] / nvalid) : 0.f;            // frac_predicted_win
        a.out[8] = nvalid > 0 ? (float)(tot[7] / nvalid) : 0.f;            // frac_predicted_draw
        if (a.acc) {
            a.acc[0] += pl;
            a.acc[1] += a.combined_value_metric ? vs : ce;
            a.acc[2] += a.combined_value_metric ? 0.f : mse;
            a.acc[3] += ent;
        }
    }
    if (a.dvlogits) {
        const float gs = a.gscale ? *a.gscale : 1.f;
        const float wv = nvalid > 0 ? (float)(a.lambda_value / nvalid) * gs : 0.f;
        const float ws = 2.f * a.lambda_score / a.B * gs;
        for (int b = tid; b < a.B; b += 1024) {
            const long long src = a.idx ? a.idx[b] : b;
            const long long cat = a.cats[src];
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (cat >= 0) {
                const float l0 = a.vlogits[b * 3], l1 = a.vlogits[b * 3 + 1], l2 = a.vlogits[b * 3 + 2];
                const float m = fmaxf(l0, fmaxf(l1, l2));
                const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m), inv = 1.f / (e0 + e1 + e2);
                g0 = (e0 * inv - (cat == 0)) * wv; g1 = (e1 * inv - (cat == 1)) * wv; g2 = (e2 * inv - (cat == 2)) * wv;
            }
            a.dvlogits[b * 3] = g0; a.dvlogits[b * 3 + 1] = g1; a.dvlogits[b * 3 + 2] = g2;
            a.dscore[b] = ws * (a.score[b] - a.targets[src]);
        }
    }
}

// rollout side: scalar value P(W)-P(L) (+ optional blend with clamp(score,-1,1))
__global__ void scalar_value_kernel(const float* __restrict__ vl, const float* __restrict__ score, float alpha,
                                    float* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float l0 = vl[b * 3], l1 = vl[b * 3 + 1], l2 = vl[b * 3 + 2];
    const float m = fmaxf(l0, fmaxf(l1, l2));
    const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
    float v = (e0 - e2) / (e0 + e1 + e2);
    if (score && alpha != 0.f) v = (1.f - alpha) * v + alpha * fminf(fmaxf(score[b], -1.f), 1.f);
    out[b] = v;
}

// Rollout side, one launch per select_actions call (katago_ppo.py:567-612: masked_fill(-inf) -> softmax -> Categorical.sample()
// -> log_prob, zero-legal guard, scalar value): one workgroup per environment stages the logit row in LDS, draws ONE action
// by inverse-CDF over the legal actions (contiguous chunk per thread, block prefix sums, the first chunk whose running
// total passes u * total is walked) and returns log p(action) as the masked-softmax path does; the value head's
// P(W) - P(L) (+ blend) rides along.  u comes from a 64-bit mix of (seed, environment): the caller draws `seed` from the
// host generator, so torch.manual_seed() still fixes the rollout.  flags[0] |= NaN logits, flags[1] |= a row without a
// legal action (its action is 0, the caller raises as the reference does).
struct SampleArgs {
    const void* logits; int logits_bf16; const uint8_t* legal; int legal_words; unsigned long long seed;
    const float* vlogits; const float* score; float alpha;
    long long* actions; float* logp; float* values; int* nlegal; int* flags; int A;
};

__device__ __forceinline__ unsigned long long sample_mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(kPolThreads) void policy_sample_kernel(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* row = reinterpret_cast<float*>(smem);
    uint8_t* msk = reinterpret_cast<uint8_t*>(smem + ((size_t)a.A * 4 + 15) / 16 * 16);
    __shared__ float red[kPolThreads / 64];
    __shared__ float wave_tot[kPolThreads / 64];
    __shared__ int s_first, s_last;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, A = a.A;
    const uint8_t* lm = a.legal + (size_t)b * A;
    const uint32_t* lw = reinterpret_cast<const uint32_t*>(a.legal) + (size_t)b * a.legal_words;
    float mx = -INFINITY, nlegal = 0.f;
    int nan_seen = 0;
    for (int j = tid; j < A; j += kPolThreads) {
        const float v = a.logits_bf16 ? bf2f(static_cast<const uint16_t*>(a.logits)[(size_t)b * A + j])
                                      : static_cast<const float*>(a.logits)[(size_t)b * A + j];
        const uint8_t k = a.legal_words ? (uint8_t)((lw[j >> 5] >> (j & 31)) & 1u) : lm[j];
        row[j] = v; msk[j] = k;
        nan_seen |= (v != v);
        if (k) { mx = fmaxf(mx, v); nlegal += 1.f; }
    }
    if (tid == 0) { s_first = kPolThreads; s_last = -1; }
    mx = block_reduce(mx, red, true);
    nlegal = block_reduce(nlegal, red, false);
    const float nanf_ = block_reduce((float)nan_seen, red, false);
    float s = 0.f;
    for (int j = tid; j < A; j += kPolThreads)
        if (msk[j]) s += expf(row[j] - mx);
    s = block_reduce(s, red, false);                           // the normaliser of the masked-softmax path (same order)
    // contiguous chunk per thread, inclusive prefix over the threads
    const int chunk = (A + kPolThreads - 1) / kPolThreads, j0 = tid * chunk, j1 = min(j0 + chunk, A);
    float local = 0.f;
    for (int j = j0; j < j1; ++j) if (msk[j]) local += expf(row[j] - mx);
    float incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    float base = 0.f, total = 0.f;
    for (int w = 0; w < kPolThreads / 64; ++w) { if (w < wave) base += wave_tot[w]; total += wave_tot[w]; }
    incl += base;
    const unsigned long long h = sample_mix(a.seed ^ sample_mix((unsigned long long)b + 0x5851F42D4C957F2Dull));
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);   // 24 bits: [0, 1)
    const float target = u * total;
    if (local > 0.f) {
        if (incl > target) atomicMin(&s_first, tid);
        atomicMax(&s_last, tid);
    }
    __syncthreads();
    const int winner = s_first < kPolThreads ? s_first : s_last;          // (rounding can leave target >= total: last legal chunk)
    if (tid == 0) {
        if (nanf_ > 0.f) atomicOr(&a.flags[0], 1);
        if (nlegal == 0.f) { atomicOr(&a.flags[1], 1); a.actions[b] = 0; a.logp[b] = 0.f; }
        a.nlegal[b] = (int)nlegal;
        if (a.values) {                                        // katago_ppo.py:536-541 / value_adapter.py:56-65
            const float l0 = a.vlogits[b * 3], l1 = a.vlogits[b * 3 + 1], l2 = a.vlogits[b * 3 + 2];
            const float m = fmaxf(l0, fmaxf(l1, l2));
            const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
            float v = (e0 - e2) / (e0 + e1 + e2);
            if (a.score && a.alpha != 0.f) v = (1.f - a.alpha) * v + a.alpha * fminf(fmaxf(a.score[b], -1.f), 1.f);
            a.values[b] = v;
        }
    }
    if (tid == winner) {
        float run = incl - local;
        int pick = -1, last = -1;
        for (int j = j0; j < j1; ++j) {
            if (!msk[j]) continue;
            last = j;
            run += expf(row[j] - mx);
            if (run > target) { pick = j; break; }
        }
        if (pick < 0) pick = last;
        a.actions[b] = pick;
        // Categorical(probs).log_prob (katago_ppo.py:603-605) = log(clamp(p, eps, 1 - eps)), eps = FLT_EPSILON: an action
        // more than ~88 nats below the maximum gives log(eps), not -inf
        const float p = expf(row[pick] - mx) * (1.f / s);
        a.logp[b] = logf(fminf(fmaxf(p, 1.1920929e-7f), 1.f - 1.1920929e-7f));
    }
}

}  // namespace

extern "C" int ka_policy_loss(const float* logits, const void* legal, const long long* actions, const float* old_lp,
                              const float* adv, const long long* idx, float* dlogits, float* new_lp, float* rowloss,
                              float* rowent, int* flags, const float* gscale, float clip_eps, float w_policy,
                              float w_entropy, int B, int A, int legal_words, void* stream) {
    KA_REQUIRE(logits && legal && actions && old_lp && adv && new_lp && rowloss && rowent && flags && B > 0 && A > 0,
               "policy_loss: null tensor");
    KA_REQUIRE(legal_words == 0 || legal_words == (A + 31) / 32, "policy_loss: packed mask rows must hold %d words", (A + 31) / 32);
    PolicyArgs a{logits, static_cast<const uint8_t*>(legal), actions, old_lp, adv, idx, dlogits, new_lp, rowloss,
                 rowent, flags, gscale, clip_eps, w_policy, w_entropy, A, legal_words};
    const size_t lds = ((size_t)A * 4 + 15) / 16 * 16 + ((size_t)A + 15) / 16 * 16;
    KA_REQUIRE(lds <= 64 * 1024, "policy_loss: action space %d too large for the LDS row", A);
    hipLaunchKernelGGL(policy_loss_kernel, dim3(B), dim3(kPolThreads), lds, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("policy_loss");
}

extern "C" int ka_masked_softmax(const float* logits, const void* legal, float* probs, int* nlegal, int* flags, int B, int A,
                                 int legal_words, void* stream) {
    KA_REQUIRE(logits && legal && probs && nlegal && flags && B > 0 && A > 0, "masked_softmax: null tensor");
    KA_REQUIRE(legal_words == 0 || legal_words == (A + 31) / 32, "masked_softmax: packed mask rows must hold %d words", (A + 31) / 32);
    const size_t lds = ((size_t)A * 4 + 15) / 16 * 16 + ((size_t)A + 15) / 16 * 16;
    KA_REQUIRE(lds <= 64 * 1024, "masked_softmax: action space %d too large for the LDS row", A);
    hipLaunchKernelGGL(masked_softmax_kernel, dim3(B), dim3(kPolThreads), lds, static_cast<hipStream_t>(stream), logits,
                       static_cast<const uint8_t*>(legal), probs, nlegal, flags, A, legal_words);
    return ka_check_launch("masked_softmax");
}

extern "C" int ka_policy_sample(const void* logits, int logits_bf16, const void* legal, int legal_words, long long seed,
                                const float* vlogits, const float* score, float alpha, long long* actions, float* logp,
                                float* values, int* nlegal, int* flags, int B, int A, void* stream) {
    KA_REQUIRE(logits && legal && actions && logp && nlegal && flags && B > 0 && A > 0, "policy_sample: null tensor");
    KA_REQUIRE(legal_words == 0 || legal_words == (A + 31) / 32, "policy_sample: packed mask rows must hold %d words", (A + 31) / 32);
    KA_REQUIRE((values == nullptr) || vlogits, "policy_sample: values need the value logits");
    const size_t lds = ((size_t)A * 4 + 15) / 16 * 16 + ((size_t)A + 15) / 16 * 16;
    KA_REQUIRE(lds <= 64 * 1024, "policy_sample: action space %d too large for the LDS row", A);
    SampleArgs a{logits, logits_bf16, static_cast<const uint8_t*>(legal), legal_words, (unsigned long long)seed, vlogits, score, alpha,
                 actions, logp, values, nlegal, flags, A};
    hipLaunchKernelGGL(policy_sample_kernel, dim3(B), dim3(kPolThreads), lds, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("policy_sample");
}

extern "C" int ka_policy_ce(const float* logits, const long long* targets, const long long* idx, float* dlogits,
                            float* rowloss, int* flags, const float* gscale, float w_policy, int B, int A, void* stream) {
    KA_REQUIRE(logits && targets && rowloss && flags && B > 0 && A > 0, "policy_ce: null tensor");
    PolicyCeArgs a{logits, targets, idx, dlogits, rowloss, flags, gscale, w_policy, A};
    const size_t lds = ((size_t)A * 4 + 15) / 16 * 16;
    KA_REQUIRE(lds <= 64 * 1024, "policy_ce: action space %d too large for the LDS row", A);
    hipLaunchKernelGGL(policy_ce_kernel, dim3(B), dim3(kPolThreads), lds, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("policy_ce");
}

// out[9]: policy_loss, value_ce, score_mse, entropy, total, n_valid, value_accuracy, frac_win, frac_draw
extern "C" int ka_value_loss(const float* vlogits, const float* score, const long long* cats, const float* targets,
                             const long long* idx, const float* rowloss, const float* rowent, float* dvlogits,
                             float* dscore, float* out, float* acc, const float* gscale, float lambda_policy,
                             float lambda_value, float lambda_score, float entropy_coeff, int combined_value_metric,
                             int B, void* stream) {
    KA_REQUIRE(vlogits && score && cats && targets && rowloss && rowent && out && B > 0, "value_loss: null tensor");
    KA_REQUIRE((dvlogits == nullptr) == (dscore == nullptr), "value_loss: dvlogits/dscore must come together");
    ValueArgs a{vlogits, score, cats, targets, idx, rowloss, rowent, dvlogits, dscore, out, acc, gscale,
                lambda_policy, lambda_value, lambda_score, entropy_coeff, B, combined_value_metric};
    hipLaunchKernelGGL(value_loss_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("value_loss");
}

extern "C" int ka_scalar_value(const float* vlogits, const float* score, float alpha, float* out, int B, void* stream) {
    KA_REQUIRE(vlogits && out && B > 0, "scalar_value: null tensor");
    hipLaunchKernelGGL(scalar_value_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       vlogits, score, alpha, out, B);
    return ka_check_launch("scalar_value");
}
