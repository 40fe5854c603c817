// Arguments of the all-resident SampleRNN kernel (see srnn_resident.hip): every tier of the network, the bottom tier and the
// head in ONE launch per generate block, workgroup roles by blockIdx.
#pragma once
#include "mmk_common.h"

namespace mmk {

constexpr int kResMaxTiers = MMK_MAX_TIERS - 1;     // recurrent tiers (the last entry of frame_sizes is the bottom tier)

// One recurrent tier (SampleRNNTier.forward, sample_rnn_v2.py:83-99) as the kernel sees it.  Its workgroups own 16 hidden units x
// 16 MT clips each; block0 is the first blockIdx of the role.
struct SrnnResTier {
  int32_t fs;                 // frame size: the tier updates at every t % fs == 0
  int32_t up;                 // slots of its up-sampler, fs / (frame size of the tier below), or fs for the last recurrent tier
  int32_t up_mod;             // slots of the tier ABOVE per update of that tier (0: top tier); its frame size is fs * up_mod
  int32_t n_tiles;            // 16-row output tiles per workgroup: G (up - 1) link tiles of the tier below (gate rows of its slots 1 .. up - 1) - or,
                              // for the last recurrent tier, the tiles of the rows composed with the head's first layer (rpb rows per unit block)
  int32_t rpb;                // last recurrent tier: composed rows per unit block, row g = ub rpb + r -> (slot 1 + g / Hm, hidden unit g % Hm)
  int32_t block0;
  int32_t mt;                 // row tiles of 16 clips per workgroup: 1, 2 - or 4, top tier only
  int32_t w_tile_chunks;      // K-chunks per packed gate tile (H / 16 for separate matrices, 2 H / 16 for the LSTM's [x | h])
  int32_t fsp;                // fs rounded up to 4: row length of v_full
  const float* whh_wp;        // packed (linear.hip) recurrent gate matrix, tiles g KC + ub
  const float* link_wp;       // tiers with a tier above: packed W_ih W_up,above, rows j G H + g H + u (slot j of the tier above), K = H; the tier itself
                              // multiplies slot 0 (tiles g KC + ub, in registers), the tier above streams the others (its out_wp)
  const float* gconst;        // [2][G H]: the constant of the gates' input half - top tier: W_ih b_in + b_ih, else W_ih (b_in + b_up,0) + b_ih (slot 0 of
                              // the link); LSTM: + b_hh | GRU: b_hh
  const float* v_full;        // [G H][fsp]: W_ih W_in (fp64, rounded once), zero padded
  const float* out_wp;        // packed output tiles: link_wp of the tier below - last recurrent tier: W0 W_up[slot] (tile ub n_tiles + i)
  const float* out_bias;      // [up][G H] constants of the tier below's link rows, W_ih,below (b_in,below + b_up[slot]) + b_ih,below (LSTM: + b_hh) -
                              // last recurrent tier: null, the constant sits in the bottom role's table
  float* h_ring;              // [2][Bmax][H]: slot (cnt & 1) holds the state at the start, every update writes the other one
  int64_t h_slot_stride;
  float* c;                   // LSTM cell state (Bmax, H), read at the start, written at the end
  int64_t* cnt;               // update counter of the tier (srnn_gru.hip keeps the same one)
  unsigned long long* h_gran;        // [2][B][H] granules {update number, new state}, parity = update number & 1
  unsigned long long* out_gran;      // [B][up][G H] granules {update number, gate row of the tier below} (slot 0 unused) - last recurrent tier:
                                     // [B][S][Hm] {update number, W0-composed row}
  const unsigned long long* upper_gran;     // out_gran of the tier above, or null
  const unsigned long long* upper_h_gran;   // h_gran of the tier above, or null
  unsigned long long* prog;                 // teacher-forced launches: updates whose input half this tier's workgroups have read, summed over the workgroups
                                            // (zero at the start of a launch) - what paces the tier above where no drawn class does
};

struct SrnnResArgs {
  int32_t B, H, n_tiers, lstm;       // clips, hidden, recurrent tiers, rnn kind
  int32_t n_steps;
  int32_t teacher;                   // 1: the warm-up - the tiers take their windows from idx (positions + shift) instead of the bottom role's classes, the bottom
                                     // role does not run (sample_rnn_v2.py:229-234: generate_step over the prompt, its outputs dropped); the tiers pace
                                     // each other by `prog`
  int64_t shift;                     // teacher: the window of step t starts at position t - fs + shift (the prompt's offset, :230)
  int64_t t_begin;                   // first step of the block, a multiple of frame_sizes[0]
  float class_size;
  SrnnResTier tier[kResMaxTiers];
  // bottom tier + head: one workgroup per clip, blockIdx < B (sample_rnn_v2.py:252-260, networks/mlp.py, modules/targets.py)
  int32_t Hm, Q, n_out, learn_temp, fsb, S;      // MLP hidden, classes, classes + temperature column, bottom frame size, frame_sizes[-2]
  float min_temp;
  int64_t* idx; int64_t idx_rs;      // (B, T) classes, written in place
  const float* cp0;                  // (Hm, H) row-major: W0 W_up[slot 0] of the last recurrent tier (fp64, rounded once)
  const float* a_comp;               // (fsb, Hm): W0 wb_i
  const float* bcs;                  // (S, Hm): W0 (b_up[slot] + bb) + b0
  const float* fc2_raw; const float* fc2_bias;   // (n_out, Hm) row-major as bound
  const float* temperature; const float* uniforms; int64_t uni_ld, uni_off;
  float* logits_out; int64_t logits_ld;          // logits of the block's last step
  unsigned long long* cls_gran;      // [B][256] granules {position + 1, class}
  int* err;                          // sticky error word: a wait that timed out (6 bottom role, 7 tier role)
  unsigned long long* stamps;        // diagnostic: [role][8] phase totals in 100 MHz ticks, or null
};

bool srnn_resident_supported(int H, bool lstm, int Hm, int n_out, int Q, int fsb, int S);
// workgroups of the launch for B clips (0: the geometry does not fit the chip); mt_out[tier]: clips per workgroup of that tier / 16
int srnn_resident_grid(int H, int B, int n_tiers, int spare_cus, int* mt_out);
size_t srnn_resident_lds_bytes(const SrnnResArgs& a);
int launch_srnn_resident(const SrnnResArgs& a, hipStream_t stream);

}  // namespace mmk
