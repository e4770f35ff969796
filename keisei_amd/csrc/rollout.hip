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

}  // namespace

extern "C" int ka_mask_words(int A) { return (A + 31) / 32; }

extern "C" int ka_rollout_append(const float* obs, const void* legal, const long long* actions, const float* log_probs,
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
                 flags, obs_elems, A, (A + 31) / 32};
    hipLaunchKernelGGL(rollout_append_kernel, dim3(n), dim3(kAppendThreads), 0, static_cast<hipStream_t>(stream), a);
    return ka_check_launch("rollout_append");
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
