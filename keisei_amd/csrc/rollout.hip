// Device-resident rollout store: the reference's KataGoRolloutBuffer.add() pulls ten tensors to the
// host every step (katago_ppo.py:226-242, ten synchronising .cpu() calls) and update() pins and re-uploads the
// epoch (~450 MB at 64 k samples, :784-790).  Here one launch per step appends the transitions to preallocated
// HBM columns, packs the 11 259-byte legal mask of each transition into 352 words (bit j of word w = action
// 32 w + j) and evaluates the reference's four input guards (:244-266) as device flags, so the host reads back
// 16 bytes instead of the tensors.  The minibatch loss kernel (loss.hip) consumes the packed rows directly.
//
// Layout of one transition: obs row (obs_elems fp32, NCHW as the environment hands it over), mask row (`words`
// uint32), and one element in each scalar column (actions / categories / env ids int64, log-probs / values /
// rewards / score targets / override fp32, dones / terminated bytes).
#include "common.h"

namespace {

struct AppendArgs {
    // one timestep, n rows (source)
    const float* obs; const uint8_t* legal; const long long* actions; const float* log_probs; const float* values;
    const float* rewards; const uint8_t* dones; const uint8_t* terminated; const long long* cats; const float* score;
    const long long* env_ids; const float* override_;
    // columns of the store, already offset to the first row written
    float* d_obs; uint32_t* d_bits; long long* d_actions; float* d_log_probs; float* d_values; float* d_rewards;
    uint8_t* d_dones; uint8_t* d_terminated; long long* d_cats; float* d_score; long long* d_env_ids; float* d_override;
    int* flags;      // [0] terminated without done, [1] category outside {-1,0,1,2}, [2] NaN score target, [3] bits of max |score|
    int obs_elems, A, words;
    int packed;      // legal holds packed rows (n, words) uint32 already (the device env's / PendingTransitions' form): copied
};

constexpr int kAppendThreads = 256;

__global__ __launch_bounds__(kAppendThreads) void rollout_append_kernel(AppendArgs a) {
    const int row = blockIdx.x, tid = threadIdx.x;
    // observation row: 8-byte pieces when the row length is even (50*81 = 4050 floats: rows are 8- not 16-byte aligned)
    const float* so = a.obs + (size_t)row * a.obs_elems;
    float* dob = a.d_obs + (size_t)row * a.obs_elems;
    if ((a.obs_elems & 1) == 0) {
        const f32x2* s2 = reinterpret_cast<const f32x2*>(so);
        f32x2* d2 = reinterpret_cast<f32x2*>(dob);
        for (int i = tid; i < a.obs_elems / 2; i += kAppendThreads) d2[i] = s2[i];
    } else {
        for (int i = tid; i < a.obs_elems; i += kAppendThreads) dob[i] = so[i];
    }
    // legal mask: a wave reads 64 consecutive bytes and votes -> two words per ballot
    const uint8_t* lm = a.legal + (size_t)row * a.A;
    uint32_t* bits = a.d_bits + (size_t)row * a.words;
    const int lane = tid & 63, wave = tid >> 6;
    if (a.packed) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.legal) + (size_t)row * a.words;
        for (int i = tid; i < a.words; i += kAppendThreads) bits[i] = src[i];
    } else
    for (int base = wave * 64; base < a.words * 32; base += (kAppendThreads / 64) * 64) {
        const int j = base + lane;
        const unsigned long long vote = __ballot(j < a.A && lm[j] != 0);
        if (lane < 2 && (base >> 5) + lane < a.words) bits[(base >> 5) + lane] = (uint32_t)(vote >> (32 * lane));
    }
    if (tid == 0) {
        const uint8_t dn = a.dones[row] != 0, tm = a.terminated[row] != 0;
        const long long cat = a.cats[row];
        const float sc = a.score[row];
        a.d_actions[row] = a.actions[row];
        a.d_log_probs[row] = a.log_probs[row];
        a.d_values[row] = a.values[row];
        a.d_rewards[row] = a.rewards[row];
        a.d_dones[row] = dn; a.d_terminated[row] = tm;
        a.d_cats[row] = cat;
        a.d_score[row] = sc;
        if (a.d_env_ids) a.d_env_ids[row] = a.env_ids[row];
        if (a.d_override) a.d_override[row] = a.override_ ? a.override_[row] : __uint_as_float(0x7fc00000u);
        if (tm && !dn) atomicOr(&a.flags[0], 1);
        if (cat < -1 || cat > 2) atomicOr(&a.flags[1], 1);
        if (sc != sc) atomicOr(&a.flags[2], 1);
        else atomicMax(reinterpret_cast<unsigned int*>(&a.flags[3]), __float_as_uint(fabsf(sc)));
    }
}

// bool rows (rows x A bytes) from packed rows; row r of the output reads packed row idx[r] (idx NULL = identity)
__global__ __launch_bounds__(256) void unpack_mask_kernel(const uint32_t* __restrict__ bits, const long long* __restrict__ idx,
                                                          uint8_t* __restrict__ out, int A, int words) {
    const size_t row = blockIdx.x;
    const uint32_t* src = bits + (size_t)(idx ? idx[row] : (long long)row) * words;
    uint8_t* dst = out + row * A;
    for (int j = threadIdx.x; j < A; j += 256) dst[j] = (src[j >> 5] >> (j & 31)) & 1u;
}

// packed rows from bool rows (the inverse; tests and callers that hold bool masks on the device)
__global__ __launch_bounds__(256) void pack_mask_kernel(const uint8_t* __restrict__ legal, uint32_t* __restrict__ bits, int A,
                                                        int words) {
    const uint8_t* lm = legal + (size_t)blockIdx.x * A;
    uint32_t* out = bits + (size_t)blockIdx.x * words;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = wave * 64; base < words * 32; base += 256) {
        const int j = base + lane;
        const unsigned long long vote = __ballot(j < A && lm[j] != 0);
        if (lane < 2 && (base >> 5) + lane < words) out[(base >> 5) + lane] = (uint32_t)(vote >> (32 * lane));
    }
}

// ---------------------------------------------------------------- pending learner transitions
// The split-merge rollout keeps, per game, the learner's last move until its outcome is known (the opponent may move in
// between): reference keisei/training/katago_loop.py:139-250 holds eight (num_envs, ...) tensors and moves rows with
// boolean-mask indexing (a dozen launches and a synchronising nonzero() per call).  Here the slots are columns in HBM --
// observation rows, legal masks as PACKED rows (352 words instead of 11 259 bytes), scalars -- and each protocol step is one
// launch: open (scatter by game mask, masks packed or copied on the way, "slot already taken" as a device flag), settle
// (reward accumulation, selection, stable compaction in game order, value categories, slot release).
struct PendingArgs {
    // slots
    float* obs; uint32_t* bits; long long* actions; float* log_probs; float* values; float* rewards; float* score; uint8_t* valid;
    uint8_t* valid_out;   // settle: the slots' state after the launch (other workgroups still count over `valid` meanwhile)
    // open: this step's tensors (all num_envs rows) and the games to open
    const uint8_t* env_mask; const float* s_obs; const uint8_t* s_legal; const uint32_t* s_bits; const long long* s_actions;
    const float* s_log_probs; const float* s_values; const float* s_rewards; const float* s_score;
    // settle: selection, this step's flags, optional rewards to add first; compacted outputs (num_envs rows allocated)
    const uint8_t* fin_mask; const void* dones; const void* terminated; const float* add_rewards;
    float* o_obs; uint32_t* o_bits; long long* o_actions; float* o_log_probs; float* o_values; float* o_rewards; float* o_dones;
    float* o_terminated; float* o_score; long long* o_env_ids; long long* o_cats;
    int* flags;          // open: [0] a selected slot was still valid;  settle: [1] number of rows written
    int n, obs_elems, A, words, flag_f32;      // flag_f32: dones / terminated are float rows (the loop's tensors), else bytes
};

__device__ __forceinline__ bool pending_flag(const void* p, int i, int f32) {
    return f32 ? static_cast<const float*>(p)[i] != 0.f : static_cast<const uint8_t*>(p)[i] != 0;
}

// conflict probe: one workgroup; flags[0] = 1 when a game to open still holds a pending transition (reference: RuntimeError
// before anything is written, katago_loop.py:187-191 -- the open kernel reads the flag and then writes nothing)
__global__ __launch_bounds__(256) void pending_probe_kernel(PendingArgs a) {
    bool hit = false;
    for (int i = threadIdx.x; i < a.n; i += 256) hit |= a.env_mask[i] && a.valid[i];
    const int any = __syncthreads_or(hit);
    if (threadIdx.x == 0) a.flags[0] = any ? 1 : 0;
}

__global__ __launch_bounds__(256) void pending_open_kernel(PendingArgs a) {
    const int g = blockIdx.x, tid = threadIdx.x;
    if (!a.env_mask[g] || a.flags[0]) return;
    const f32x2* s2 = reinterpret_cast<const f32x2*>(a.s_obs + (size_t)g * a.obs_elems);
    f32x2* d2 = reinterpret_cast<f32x2*>(a.obs + (size_t)g * a.obs_elems);
    if ((a.obs_elems & 1) == 0) { for (int i = tid; i < a.obs_elems / 2; i += 256) d2[i] = s2[i]; }
    else { for (int i = tid; i < a.obs_elems; i += 256) a.obs[(size_t)g * a.obs_elems + i] = a.s_obs[(size_t)g * a.obs_elems + i]; }
    uint32_t* bits = a.bits + (size_t)g * a.words;
    if (a.s_bits) {
        for (int w = tid; w < a.words; w += 256) bits[w] = a.s_bits[(size_t)g * a.words + w];
    } else {
        const uint8_t* lm = a.s_legal + (size_t)g * a.A;
        const int lane = tid & 63, wave = tid >> 6;
        for (int base = wave * 64; base < a.words * 32; base += 256) {
            const int j = base + lane;
            const unsigned long long vote = __ballot(j < a.A && lm[j] != 0);
            if (lane < 2 && (base >> 5) + lane < a.words) bits[(base >> 5) + lane] = (uint32_t)(vote >> (32 * lane));
        }
    }
    if (tid == 0) {
        a.actions[g] = a.s_actions[g]; a.log_probs[g] = a.s_log_probs[g]; a.values[g] = a.s_values[g];
        a.rewards[g] = a.s_rewards[g]; a.score[g] = a.s_score[g]; a.valid[g] = 1;
    }
}

// rewards[valid] += add[valid] (katago_loop.py:203-211) -- also available fused into the settle launch
__global__ void pending_accumulate_kernel(PendingArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n && a.valid[i]) a.rewards[i] += a.add_rewards[i];
}

// settle: one workgroup per game.  Row of game g in the output = number of selected games before it (every workgroup counts
// the prefix itself: num_envs bytes, L2-resident), so the rows come out in game order like the reference's nonzero().
__global__ __launch_bounds__(256) void pending_settle_kernel(PendingArgs a) {
    const int g = blockIdx.x, tid = threadIdx.x;
    __shared__ int s_cnt[4];
    int before = 0, total = 0;
    for (int i = tid; i < a.n; i += 256) {
        const int sel = a.fin_mask[i] && a.valid[i];
        total += sel; before += sel && i < g;
    }
    // (two sums packed in one reduction: counts stay below 2^15 for any plausible num_envs; checked by the launcher)
    int packed = before | (total << 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) packed += __shfl_xor(packed, o);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = packed;
    __syncthreads();
    packed = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const int row = packed & 0xffff, ntot = packed >> 16;
    if (g == 0 && tid == 0) a.flags[1] = ntot;
    const bool mine = a.valid[g] != 0;
    float rw = mine ? a.rewards[g] : 0.f;
    if (mine && a.add_rewards) rw += a.add_rewards[g];
    if (!(a.fin_mask[g] && mine)) {
        if (tid == 0) {
            a.valid_out[g] = mine;
            if (mine && a.add_rewards) a.rewards[g] = rw;              // stays pending: keeps the accumulated reward
        }
        return;
    }
    const f32x2* s2 = reinterpret_cast<const f32x2*>(a.obs + (size_t)g * a.obs_elems);
    f32x2* d2 = reinterpret_cast<f32x2*>(a.o_obs + (size_t)row * a.obs_elems);
    if ((a.obs_elems & 1) == 0) { for (int i = tid; i < a.obs_elems / 2; i += 256) d2[i] = s2[i]; }
    else { for (int i = tid; i < a.obs_elems; i += 256) a.o_obs[(size_t)row * a.obs_elems + i] = a.obs[(size_t)g * a.obs_elems + i]; }
    for (int w = tid; w < a.words; w += 256) a.o_bits[(size_t)row * a.words + w] = a.bits[(size_t)g * a.words + w];
    if (tid == 0) {
        const bool dn = pending_flag(a.dones, g, a.flag_f32), tm = pending_flag(a.terminated, g, a.flag_f32);
        a.o_actions[row] = a.actions[g]; a.o_log_probs[row] = a.log_probs[g]; a.o_values[row] = a.values[g];
        a.o_rewards[row] = rw; a.o_score[row] = a.score[g];
        a.o_dones[row] = dn ? 1.f : 0.f; a.o_terminated[row] = tm ? 1.f : 0.f;
        a.o_env_ids[row] = g;
        // value-head label of the settled transition (katago_loop.py:75-92): only genuinely terminal positions get one
        a.o_cats[row] = !tm ? -1 : (rw > 0.f ? 0 : (rw == 0.f ? 1 : 2));
        a.valid_out[g] = 0; a.rewards[g] = 0.f;
    }
}

}  // namespace

extern "C" int ka_mask_words(int A) { return (A + 31) / 32; }

static int rollout_append_impl(const float* obs, const void* legal, int packed, const long long* actions, const float* log_probs,
                                 const float* values, const float* rewards, const void* dones, const void* terminated,
                                 const long long* cats, const float* score, const long long* env_ids, const float* override_,
                                 float* d_obs, void* d_bits, long long* d_actions, float* d_log_probs, float* d_values,
                                 float* d_rewards, void* d_dones, void* d_terminated, long long* d_cats, float* d_score,
                                 long long* d_env_ids, float* d_override, int* flags, int n, int obs_elems, int A,
                                 void* stream) {
    KA_REQUIRE(obs && legal && actions && log_probs && values && rewards && dones && terminated && cats && score,
               "rollout_append: null source column");
    KA_REQUIRE(d_obs && d_bits && d_actions && d_log_probs && d_values && d_rewards && d_dones && d_terminated && d_cats &&
               d_score && flags, "rollout_append: null store column");
    KA_REQUIRE(n > 0 && obs_elems > 0 && A > 0, "rollout_append: bad sizes (n=%d obs=%d A=%d)", n, obs_elems, A);
    KA_REQUIRE(!d_env_ids || env_ids, "rollout_append: the store has an env_ids column but none was supplied");
    AppendArgs a{obs, static_cast<const uint8_t*>(legal), actions, log_probs, values, rewards,
                 static_cast<const uint8_t*>(dones), static_cast<const uint8_t*>(terminated), cats, score, env_ids, override_,
                 d_obs, static_cast<uint32_t*>(d_bits), d_actions, d_log_probs, d_values, d_rewards,
                 static_cast<uint8_t*>(d_dones), static_cast<uint8_t*>(d_terminated), d_cats, d_score, d_env_ids, d_override,
                 flags, obs_elems, A, (A + 31) / 32, packed};
    hipLaunchKernelGGL(rollout_append_kernel, dim3(n), dim3(kAppendThreads), 0, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("rollout_append");
}
extern "C" int ka_rollout_append(const float* obs, const void* legal, const long long* actions, const float* log_probs,
                                 const float* values, const float* rewards, const void* dones, const void* terminated,
                                 const long long* cats, const float* score, const long long* env_ids, const float* override_,
                                 float* d_obs, void* d_bits, long long* d_actions, float* d_log_probs, float* d_values,
                                 float* d_rewards, void* d_dones, void* d_terminated, long long* d_cats, float* d_score,
                                 long long* d_env_ids, float* d_override, int* flags, int n, int obs_elems, int A,
                                 void* stream) {
    return rollout_append_impl(obs, legal, 0, actions, log_probs, values, rewards, dones, terminated, cats, score, env_ids, override_,
                               d_obs, d_bits, d_actions, d_log_probs, d_values, d_rewards, d_dones, d_terminated, d_cats, d_score,
                               d_env_ids, d_override, flags, n, obs_elems, A, stream);
}
// the same with the legal masks handed over as PACKED rows (n, ka_mask_words(A)) uint32 -- what the device env and
// PendingTransitions.finalize() hold: the words are copied, nothing is unpacked and packed again on the way
extern "C" int ka_rollout_append_packed(const float* obs, const void* legal_bits, const long long* actions, const float* log_probs,
                                        const float* values, const float* rewards, const void* dones, const void* terminated,
                                        const long long* cats, const float* score, const long long* env_ids, const float* override_,
                                        float* d_obs, void* d_bits, long long* d_actions, float* d_log_probs, float* d_values,
                                        float* d_rewards, void* d_dones, void* d_terminated, long long* d_cats, float* d_score,
                                        long long* d_env_ids, float* d_override, int* flags, int n, int obs_elems, int A,
                                        void* stream) {
    return rollout_append_impl(obs, legal_bits, 1, actions, log_probs, values, rewards, dones, terminated, cats, score, env_ids, override_,
                               d_obs, d_bits, d_actions, d_log_probs, d_values, d_rewards, d_dones, d_terminated, d_cats, d_score,
                               d_env_ids, d_override, flags, n, obs_elems, A, stream);
}

extern "C" int ka_unpack_mask_bits(const void* bits, const long long* idx, void* out, int rows, int A, void* stream) {
    KA_REQUIRE(bits && out && rows > 0 && A > 0, "unpack_mask_bits: bad arguments");
    hipLaunchKernelGGL(unpack_mask_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint32_t*>(bits), idx, static_cast<uint8_t*>(out), A, (A + 31) / 32);
    return ka_check_launch("unpack_mask_bits");
}

extern "C" int ka_pack_mask_bits(const void* legal, void* bits, int rows, int A, void* stream) {
    KA_REQUIRE(legal && bits && rows > 0 && A > 0, "pack_mask_bits: bad arguments");
    hipLaunchKernelGGL(pack_mask_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint8_t*>(legal), static_cast<uint32_t*>(bits), A, (A + 31) / 32);
    return ka_check_launch("pack_mask_bits");
}

// ---- pending learner transitions (keisei_amd/training/katago_loop.py PendingTransitions; reference katago_loop.py:139-250)
extern "C" int ka_pending_open(float* obs, void* bits, long long* actions, float* log_probs, float* values, float* rewards,
                               float* score, void* valid, const void* env_mask, const float* s_obs, const void* s_legal,
                               const void* s_bits, const long long* s_actions, const float* s_log_probs, const float* s_values,
                               const float* s_rewards, const float* s_score, int* flags, int n, int obs_elems, int A,
                               void* stream) {
    KA_REQUIRE(obs && bits && actions && log_probs && values && rewards && score && valid, "pending_open: null slot column");
    KA_REQUIRE(env_mask && s_obs && (s_legal || s_bits) && s_actions && s_log_probs && s_values && s_rewards && s_score && flags,
               "pending_open: null source");
    KA_REQUIRE(n > 0 && obs_elems > 0 && A > 0, "pending_open: bad sizes");
    PendingArgs a{};
    a.obs = obs; a.bits = static_cast<uint32_t*>(bits); a.actions = actions; a.log_probs = log_probs; a.values = values;
    a.rewards = rewards; a.score = score; a.valid = static_cast<uint8_t*>(valid);
    a.env_mask = static_cast<const uint8_t*>(env_mask); a.s_obs = s_obs; a.s_legal = static_cast<const uint8_t*>(s_legal);
    a.s_bits = static_cast<const uint32_t*>(s_bits); a.s_actions = s_actions; a.s_log_probs = s_log_probs; a.s_values = s_values;
    a.s_rewards = s_rewards; a.s_score = s_score; a.flags = flags;
    a.n = n; a.obs_elems = obs_elems; a.A = A; a.words = (A + 31) / 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pending_probe_kernel, dim3(1), dim3(256), 0, st, a);
    hipLaunchKernelGGL(pending_open_kernel, dim3(n), dim3(256), 0, st, a);
    return ka_check_launch("pending_open");
}

extern "C" int ka_pending_accumulate(float* rewards, const void* valid, const float* add_rewards, int n, void* stream) {
    KA_REQUIRE(rewards && valid && add_rewards && n > 0, "pending_accumulate: bad arguments");
    PendingArgs a{};
    a.rewards = rewards; a.valid = const_cast<uint8_t*>(static_cast<const uint8_t*>(valid)); a.add_rewards = add_rewards; a.n = n;
    hipLaunchKernelGGL(pending_accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("pending_accumulate");
}

extern "C" int ka_pending_settle(float* obs, void* bits, long long* actions, float* log_probs, float* values, float* rewards,
                                 float* score, const void* valid, void* valid_out, const void* fin_mask, const void* dones, const void* terminated,
                                 int flags_are_f32, const float* add_rewards, float* o_obs, void* o_bits, long long* o_actions,
                                 float* o_log_probs, float* o_values, float* o_rewards, float* o_dones, float* o_terminated,
                                 float* o_score, long long* o_env_ids, long long* o_cats, int* flags, int n, int obs_elems, int A,
                                 void* stream) {
    KA_REQUIRE(obs && bits && actions && log_probs && values && rewards && score && valid && valid_out && valid != valid_out,
               "pending_settle: null slot column (valid and valid_out must be two buffers)");
    KA_REQUIRE(fin_mask && dones && terminated && flags, "pending_settle: null selection");
    KA_REQUIRE(o_obs && o_bits && o_actions && o_log_probs && o_values && o_rewards && o_dones && o_terminated && o_score &&
               o_env_ids && o_cats, "pending_settle: null output column");
    KA_REQUIRE(n > 0 && n < 32768 && obs_elems > 0 && A > 0, "pending_settle: bad sizes (n=%d: at most 32767 games)", n);
    PendingArgs a{};
    a.obs = obs; a.bits = static_cast<uint32_t*>(bits); a.actions = actions; a.log_probs = log_probs; a.values = values;
    a.rewards = rewards; a.score = score; a.valid = const_cast<uint8_t*>(static_cast<const uint8_t*>(valid));
    a.valid_out = static_cast<uint8_t*>(valid_out);
    a.fin_mask = static_cast<const uint8_t*>(fin_mask); a.dones = dones; a.terminated = terminated; a.flag_f32 = flags_are_f32;
    a.add_rewards = add_rewards;
    a.o_obs = o_obs; a.o_bits = static_cast<uint32_t*>(o_bits); a.o_actions = o_actions; a.o_log_probs = o_log_probs;
    a.o_values = o_values; a.o_rewards = o_rewards; a.o_dones = o_dones; a.o_terminated = o_terminated; a.o_score = o_score;
    a.o_env_ids = o_env_ids; a.o_cats = o_cats; a.flags = flags;
    a.n = n; a.obs_elems = obs_elems; a.A = A; a.words = (A + 31) / 32;
    hipLaunchKernelGGL(pending_settle_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("pending_settle");
}
