/*
 * mmk.h — C ABI of the MI355X (gfx950) hot-path library `libmmk_hip.so`.
 *
 * Scope: the autoregressive generate path of ktonal/mimikit and its mu-law /
 * STFT feature functionals.  The reference reaches this path
 * through two *Python* protocols, not an FFI; each entry point below names the
 * reference function it stands in for (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns MMK_OK (0) or a negative error code and never
 *     throws; `mmk_last_error()` returns a thread-local message for the last
 *     failure;
 *   - all data buffers are CALLER-OWNED DEVICE pointers (e.g. torch
 *     `tensor.data_ptr()`); sizes and strides are explicit, in ELEMENTS;
 *   - the library allocates no device memory: plans take a caller-provided
 *     workspace whose size is reported by `*_workspace_bytes`;
 *   - kernels are enqueued on the `stream` argument (a `hipStream_t`, passed as
 *     void*) and the library never synchronises it, except inside
 *     `*_commit` which may wait for its own one-off graph capture;
 *   - no torch / C++ types cross the boundary;
 *   - diagnostics (in-kernel phase stamps, timing experiments that change results) are compiled only into the diagnostic
 *     build of the library (`python -m mimikit_amd.build --diag` -> libmmk_hip_diag.so, -DMMK_DIAG); the product library has
 *     no code path that produces wrong results on purpose.
 */
#ifndef MMK_H_
#define MMK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMK_ABI_VERSION 4   /* 2: exec_mode in the WaveNet / SampleRNN / Seq2Seq configs, mmk_*_sync_status for all three, mmk_*_inject_sync_error;
                              * 3: `tuning` text at the end of the three configs - the library reads no environment variable;
                              * 4: act_f / act_g in the WaveNet config, mlp_act in all three, mmk_srnn_resident_warmups */
/* activations (mimikit/modules/activations.py: ActivationEnum, the members the HIP path evaluates) */
#define MMK_ACT_IDENTITY 0
#define MMK_ACT_TANH 1
#define MMK_ACT_SIGMOID 2
#define MMK_ACT_MISH 3
#define MMK_ACT_ABS 4
#define MMK_ACT_RELU 5
#define MMK_ACT_SOFTPLUS 6
#define MMK_ACT_SIN 7
#define MMK_ACT_COS 8
#define MMK_TUNING_CHARS 256

#define MMK_OK 0
#define MMK_ERR_INVALID (-1)     /* bad argument / shape / unsupported option value */
#define MMK_ERR_HIP (-2)         /* a HIP runtime call failed */
#define MMK_ERR_UNSUPPORTED (-3) /* option combination outside this library's coverage */
#define MMK_ERR_WORKSPACE (-4)   /* workspace too small or misaligned */
#define MMK_ERR_STATE (-5)       /* call sequence violated (e.g. generate before commit) */
#define MMK_ERR_KEY (-6)         /* unknown / missing state_dict key */

#define MMK_MAX_LAYERS 128
#define MMK_MAX_COND 4
#define MMK_MAX_TIERS 8
#define MMK_MAX_MLP_HIDDEN 4
#define MMK_MAX_STREAMS 4                  /* inputs / targets of one network (len(IOSpec.inputs), len(IOSpec.targets)) */

typedef void* mmk_stream_t; /* hipStream_t */

int mmk_abi_version(void);
/* sizeof the config struct as this library was compiled: 0 mmk_wavenet_config, 1 mmk_srnn_config, 2 mmk_s2s_config; -1 for any other number -
 * what a binding that mirrors the structs by hand (ctypes, cgo, JNI) compares its own layout with before the first call */
int64_t mmk_config_bytes(int which);
/* a digest of the sources this library was compiled from (mimikit_amd/build.py: sha256 over every translation unit and header,
 * 32 hex digits; "unknown" for a build outside that script): how a caller tells a stale prebuilt library from a current one */
const char* mmk_build_digest(void);
const char* mmk_last_error(void);
/* Diagnostic: weight re-packing kernels (`*_commit`, mmk_pack_weight_f32) launched since the library was loaded.
 * The host mirror re-commits a plan only when a parameter changed (in the reference `before_generate` never
 * touches the weights, mimikit/networks/wavenet_v2.py:368-445); tests assert that through this counter. */
int64_t mmk_pack_launch_count(void);
/* Content fingerprint of `n_words` 32-bit words on the device (position-mixed hash, summed as 64-bit integers: deterministic): the host
 * mirror takes it over the concatenated weights where a generation starts, to notice writes through `tensor.data` that no version
 * counter records.  `out`: one uint64 on the device (cleared by the call). */
int mmk_fingerprint_u32(const void* words, int64_t n_words, uint64_t* out, mmk_stream_t stream);
/* the same number for `n_buffers` device buffers taken as one concatenation (host arrays of pointers and word counts), without
 * concatenating them: one launch per 96 buffers */
int mmk_fingerprint_buffers_u32(const void* const* buffers, const int64_t* n_words, int32_t n_buffers, uint64_t* out, mmk_stream_t stream);

/* ------------------------------------------------------------------------
 * Feature functionals
 * ---------------------------------------------------------------------- */

/* MuLawCompress.torch_func (mimikit/features/functionals.py:330-338).
 * codes[i] = int64(trunc((sign(x)·log1p(mu·|x|·C)/log1p(mu·C) + 1)/2·mu + 0.5)), mu = q_levels-1.
 * `edges` holds the q_levels-1 ascending fp32 decision thresholds of that
 * formula (edges[c-1] = smallest x whose code is >= c), so in-range inputs are
 * quantised exactly as the reference does; inputs outside [edges[0], +1] fall
 * back to direct fp32 evaluation of the formula (no clamp, as the reference). */
int mmk_mulaw_compress_f32_i64(const float* x, int64_t* codes, int64_t n, int32_t q_levels,
                               float compression, const float* edges, mmk_stream_t stream);

/* MuLawExpand.torch_func (mimikit/features/functionals.py:361-369).
 * `table` holds the q_levels expanded values of codes 0..q_levels-1; codes
 * outside that range are evaluated directly in fp32. */
int mmk_mulaw_expand_i64_f32(const int64_t* codes, float* x, int64_t n, int32_t q_levels,
                             float compression, const float* table, mmk_stream_t stream);

/* Resample.torch_func (mimikit/features/functionals.py:292-310) = torchaudio.functional.resample(x, orig_sr, target_sr):
 * polyphase windowed-sinc FIR.  orig / nnew are the two rates divided by their gcd; `table` is the (nnew, 2*width + orig)
 * filter bank torchaudio builds (Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99), row j = output phase j.
 * x: (batch, n_in) rows x_row_stride apart; out: (batch, mmk_resample_n_out(n_in, orig, nnew)) rows out_row_stride apart.
 * Used between the networks of an EnsembleGenerator event (mimikit/models/ensemble_generator.py:113-144). */
int64_t mmk_resample_n_out(int64_t n_in, int32_t orig, int32_t nnew);
int mmk_resample_f32(const float* x, int64_t x_row_stride, int32_t batch, int64_t n_in, const float* table, int32_t orig,
                     int32_t nnew, int32_t width, float* out, int64_t out_row_stride, mmk_stream_t stream);

/* STFT.torch_func with coordinate="mag" == MagSpec.torch_func
 * (mimikit/features/functionals.py:507-524, :576-606): periodic-Hann framed
 * real FFT magnitudes.  x: (batch, n_samples) rows `x_row_stride` apart,
 * already length-fixed by the caller (STFT._fix_length, :468-486).
 * center != 0 pads n_fft/2 zeros on both sides (pad_mode="constant").
 * out: (batch, n_frames, n_fft/2+1) contiguous, n_frames as returned by
 * mmk_stft_n_frames.  n_fft must be a power of two in [64, 4096]. */
int64_t mmk_stft_n_frames(int64_t n_samples, int32_t n_fft, int32_t hop, int32_t center);
int mmk_stft_mag_f32(const float* x, int64_t x_row_stride, int32_t batch, int64_t n_samples,
                     int32_t n_fft, int32_t hop, int32_t center, float* out, mmk_stream_t stream);

/* STFT.torch_func with a complex coordinate (mimikit/features/functionals.py:506-523):
 * torch.stft(x, n_fft, hop, window=hann_window(n_fft), center, pad_mode, return_complex=True)
 * transposed to (batch, n_frames, n_fft/2+1), then
 *   coordinate 0 'car'   -> (..., 2) = (real, imag)
 *   coordinate 1 'pol'   -> (..., 2) = (abs, angle)
 *   coordinate 2 'angle' -> angle only, no trailing dimension.
 * reflect != 0 selects pad_mode="reflect" (needs n_samples > n_fft/2), else zeros.
 * n_fft: a power of two in [64, 4096] (MMK_ERR_UNSUPPORTED otherwise). */
#define MMK_STFT_CAR 0
#define MMK_STFT_POL 1
#define MMK_STFT_ANGLE 2
int mmk_stft_f32(const float* x, int64_t x_row_stride, int32_t batch, int64_t n_samples, int32_t n_fft,
                 int32_t hop, int32_t center, int32_t reflect, int32_t coordinate, float* out,
                 mmk_stream_t stream);

/* ISTFT.torch_func (mimikit/features/functionals.py:553-564):
 * torch.istft(spec^T, n_fft, hop, window=hann_window(n_fft)) with torch's defaults (center=True,
 * length=None): inverse real FFT of every frame, periodic-Hann window, overlap-add, division by the
 * overlap-added squared window, n_fft/2 samples trimmed on both sides.
 * spec: (batch, n_frames, n_fft/2+1, 2) contiguous; coordinate 0: (real, imag), 1: (abs, angle)
 * [the reference's 'pol': abs * exp(1j * angle)].  out: (batch, hop * (n_frames - 1)).
 * work: mmk_istft_workspace_floats() floats of device scratch (0 for n_fft = 1024, where it may be NULL).
 * n_fft: a power of two in [64, 4096]; 1 <= hop < n_fft; n_frames >= 2. */
int64_t mmk_istft_n_samples(int64_t n_frames, int32_t n_fft, int32_t hop);
size_t mmk_istft_workspace_floats(int32_t batch, int64_t n_frames, int32_t n_fft);
int mmk_istft_f32(const float* spec, int32_t coordinate, int32_t batch, int64_t n_frames, int32_t n_fft,
                  int32_t hop, float* work, float* out, mmk_stream_t stream);

/* GLA.torch_func (mimikit/features/functionals.py:634-642) = torchaudio.transforms.GriffinLim(
 * n_fft, hop_length, power=1.) as published in torchaudio 2.0.1 (functional.griffinlim; the
 * reference pins that version in pyproject.toml:63-68):
 *   angles <- init;  tprev <- 0;  m = momentum / (1 + momentum)
 *   n_iter times:  inverse = istft(mag * angles);  rebuilt = stft(inverse, center, reflect)
 *                  angles = rebuilt - m * tprev;  angles /= |angles| + 1e-16;  tprev = rebuilt
 *   out = istft(mag * angles)
 * mag: (batch, n_frames, n_fft/2+1) magnitudes (time x freq, as the functional receives them).
 * init: (batch, n_frames, n_fft/2+1, 2) initial complex "angles" (torchaudio draws torch.rand of a
 * complex dtype: both parts uniform in [0, 1)), or NULL for rand_init=False (all 1 + 0i).
 * out: (batch, hop * (n_frames - 1)), which must exceed n_fft/2 (reflect padding).
 * work: mmk_gla_workspace_floats() floats of device scratch. */
size_t mmk_gla_workspace_floats(int32_t batch, int64_t n_frames, int32_t n_fft, int32_t hop);
int mmk_gla_f32(const float* mag, const float* init, int32_t batch, int64_t n_frames, int32_t n_fft,
                int32_t hop, int32_t n_iter, float momentum, float* work, float* out, mmk_stream_t stream);

/* ------------------------------------------------------------------------
 * Building blocks (exported for unit parity tests and for host-side reuse)
 * ---------------------------------------------------------------------- */

/* floats needed to hold an (N x K) weight in MFMA fragment order */
int64_t mmk_packed_weight_floats(int32_t n_rows, int32_t k_cols);
/* W: (N, K) row-major fp32 with leading dimension ldw -> packed order */
int mmk_pack_weight_f32(const float* w, int64_t ldw, int32_t n_rows, int32_t k_cols, float* packed,
                        mmk_stream_t stream);
/* Y[M,N] = act(X[M,K] @ W^T + bias); act: 0 none, 1 tanh, 2 sigmoid, 3 mish, 4 abs, 5 relu
 * (nn.Linear / 1x1 nn.Conv1d as used by modules/io.py, networks/mlp.py) */
int mmk_linear_f32(const float* x, int64_t ldx, int32_t m_rows, const float* packed_w, const float* bias,
                   int32_t n_rows, int32_t k_cols, float* y, int64_t ldy, int32_t act, mmk_stream_t stream);

/* MLP temperature column + CategoricalSampler
 * (mimikit/networks/mlp.py:58-63, mimikit/modules/targets.py:37-52).
 * logits: (rows, n_classes + has_temp_col) raw MLP outputs, `ld` apart.
 * temperature == NULL -> argmax (first maximum on ties); otherwise one
 * temperature per row and `uniforms` (one U[0,1) per row) drive inverse-CDF
 * sampling of softmax(logits / T).  out[r * out_stride] receives the class. */
int mmk_categorical_sample_f32_i64(const float* logits, int64_t ld, int32_t rows, int32_t n_classes,
                                   int32_t has_temp_col, float min_temp, const float* temperature,
                                   const float* uniforms, int64_t* out, int64_t out_stride,
                                   mmk_stream_t stream);

/* ------------------------------------------------------------------------
 * WaveNet (mimikit/networks/wavenet_v2.py)
 * ---------------------------------------------------------------------- */
typedef struct mmk_wavenet_config {
  int32_t n_layers;                        /* sum(blocks) */
  int32_t kernel_size[MMK_MAX_LAYERS];     /* get_kernels_and_dilation :295-327 */
  int32_t dilation[MMK_MAX_LAYERS];
  int32_t q_levels;                        /* class_size of input 0 (EmbeddingIO) ; 0 -> LinearIO input */
  int32_t in_dim;                          /* feature size of input 0 when q_levels == 0 */
  int32_t dim_dilated;                     /* dims_dilated[0] */
  int32_t residuals_dim;                   /* 0 = None */
  int32_t skips_dim;                       /* 0 = None */
  int32_t n_cond;                          /* len(dims_1x1) */
  int32_t cond_in_dim[MMK_MAX_COND];       /* feature size of input 1+j (LinearIO) */
  int32_t cond_dim[MMK_MAX_COND];          /* dims_1x1[j] */
  int32_t cond_q_levels[MMK_MAX_COND];     /* > 0: input 1+j is a stream of class indices through an EmbeddingIO (no bias) of that many
                                            * classes - cond[j] of the calls below is then int64 (batch, T); 0: fp32 features through a LinearIO */
  int32_t bias;                            /* Config.bias */
  int32_t gated;                           /* act_g is not None */
  int32_t act_f, act_g;                    /* Config.act_f / act_g as MMK_ACT_* codes (act_g ignored when gated == 0).  Anything but Tanh / Sigmoid runs
                                            * on the launch path (the persistent kernels and the prefill have the default gate built in) */
  int32_t head_kind;                       /* 0: MLPIO + categorical sampler, 1: linear + Abs (magspec), 2: linear */
  int32_t mlp_hidden;                      /* MLPIO.hidden_dim */
  int32_t mlp_n_hidden;                    /* MLPIO.n_hidden_layers */
  int32_t mlp_act;                         /* MLPIO.activation of every MLP head as an MMK_ACT_* code (modules/io.py:205: Mish); anything else: the launch path */
  int32_t out_dim;                         /* q_levels of the target, or n_bins */
  int32_t learn_temp;                      /* MLP.learn_temperature */
  float min_temp;
  int32_t max_batch;
  /* reverse_layer_order (:253): the layers run in reversed construction order, so the layer built without a residual
   * convolution (:216) is no longer the last one.  res_explicit != 0: layer_has_res[l] says whether layer l (in RUN
   * order) has its conv_res; 0: every layer but the last has one when residuals_dim == dim_dilated. */
  int32_t res_explicit;
  int32_t layer_has_res[MMK_MAX_LAYERS];
  int32_t layerwise_inputs;                /* Config.layerwise_inputs: the embedded input 0 is added to every layer's output (:285-286) */
  int32_t exec_mode;                       /* how the steps of a call are run: 0 = the library chooses (a persistent kernel where the
                                            * geometry and the device allow one), 1 = one fused kernel per layer half, hipGraph-replayed -
                                            * needs no co-residency of workgroups: what a caller asks for to redo a batch after
                                            * mmk_wavenet_sync_status reported a timed-out hand-off */
  int32_t with_affine_residuals;           /* Config.with_affine_residuals (:121-122, :148-149): every layer's input goes through
                                            * x_hat * a + b of a 1x1 convolution to 3 x its width (ParametrizedLinear) first; launch path,
                                            * without pad_side, layerwise_inputs, or conditioning inputs of an ungated network */
  /* more than one target (WaveNet.forward returns one output per output module, :293; the generate loop writes output k into input k,
   * loops/generate.py:213-218): n_targets in [0, MMK_MAX_STREAMS], 0 = 1.  Target 0 is the head described above and is written to in0;
   * target k >= 1 is an MLPIO + categorical sampler of x_out_dim[k] classes on the same hidden vector, written IN PLACE to cond[k - 1],
   * which must be a class stream (cond_q_levels[k - 1] > 0).  Entry 0 of the x_ arrays is unused.  Such networks run on the launch path. */
  int32_t n_targets;
  int32_t x_out_dim[MMK_MAX_STREAMS], x_mlp_hidden[MMK_MAX_STREAMS], x_mlp_n_hidden[MMK_MAX_STREAMS], x_learn_temp[MMK_MAX_STREAMS];
  float x_min_temp[MMK_MAX_STREAMS];
  char tuning[MMK_TUNING_CHARS];           /* execution switches of THIS plan as "NAME=VALUE;NAME=VALUE" (empty: the library's choices), e.g.
                                            * "MMK_WN_SPIPE=0;MMK_WN_CHAIN=1" - what the parity tests use to put one network on every kernel that can run it.  The
                                            * library reads no environment variable (the diagnostic build, -DMMK_DIAG, falls back to it) */
} mmk_wavenet_config;

typedef struct mmk_wavenet_plan mmk_wavenet_plan;

int mmk_wavenet_plan_create(const mmk_wavenet_config* cfg, mmk_wavenet_plan** out);
void mmk_wavenet_plan_destroy(mmk_wavenet_plan* plan);
/* bind one tensor of the network's state_dict by its reference key
 * (SURVEY.md section 8(a) row a4), e.g. "layers.3.conv_dil.0.0.weight". */
int mmk_wavenet_plan_bind(mmk_wavenet_plan* plan, const char* key, const float* dev_ptr, int64_t numel);
int64_t mmk_wavenet_receptive_field(const mmk_wavenet_plan* plan); /* WaveNet.rf :337-339 */
size_t mmk_wavenet_workspace_bytes(const mmk_wavenet_plan* plan);
/* packs all bound weights into `workspace` and clears the dilation queues */
int mmk_wavenet_commit(mmk_wavenet_plan* plan, void* workspace, size_t workspace_bytes, mmk_stream_t stream);
/* Teacher-forced pass over positions [t_begin, t_end) of the inputs: fills the
 * per-layer dilation queues exactly as a full-window WaveNet.forward
 * (:276-293) over those positions would see them.  in0: int64 class indices
 * (batch, T) when q_levels > 0, else fp32 (batch, T, in_dim).  cond[j]: fp32
 * (batch, T, cond_in_dim[j]), or int64 class indices (batch, T) when cond_q_levels[j] > 0.  Strides in elements. */
int mmk_wavenet_warmup(mmk_wavenet_plan* plan, int32_t batch, const void* in0, int64_t in0_row_stride,
                       const void* const* cond, const int64_t* cond_row_stride, int64_t t_begin,
                       int64_t t_end, mmk_stream_t stream);
/* n_steps of GenerateLoopV2's hot loop (mimikit/loops/generate.py:207-219)
 * fused with WaveNet.generate_step (:447-452): for t in [t0, t0+n_steps) the
 * class drawn from the network output is written IN PLACE to in0[:, t].
 * temperature: NULL (argmax) or `batch` floats; uniforms: (batch, n_steps) - with n_targets > 1 (n_targets, batch, n_steps), and the
 * class of target k >= 1 is written IN PLACE to cond[k - 1][:, t] (the const of that argument does not cover those streams). */
int mmk_wavenet_generate(mmk_wavenet_plan* plan, int32_t batch, void* in0, int64_t in0_row_stride,
                         const void* const* cond, const int64_t* cond_row_stride, int64_t t0,
                         int64_t n_steps, const float* temperature, const float* uniforms,
                         mmk_stream_t stream);
/* raw head outputs of the most recent step: (batch, out_dim + learn_temp) fp32 */
int mmk_wavenet_last_logits(mmk_wavenet_plan* plan, int32_t batch, float* out, int64_t ld, mmk_stream_t stream);
/* the same for target k: (batch, x_out_dim[k] + x_learn_temp[k]) */
int mmk_wavenet_last_logits_of(mmk_wavenet_plan* plan, int32_t target, int32_t batch, float* out, int64_t ld, mmk_stream_t stream);
/* Measurement aid for bench.py: runs n_steps like mmk_wavenet_generate (greedy) but eagerly, with HIP
 * start/stop events attached to every fused-linear launch on `stream`; waits for completion and returns
 * summed device time (ms) and launch counts per kernel class: [0] dilated+cond+gate layer kernel,
 * [1] residual+skip layer kernel, [2] input / conditioning / head linears. */
int mmk_wavenet_profile_steps(mmk_wavenet_plan* plan, int32_t batch, void* in0, int64_t in0_row_stride,
                              const void* const* cond, const int64_t* cond_row_stride, int64_t t0,
                              int64_t n_steps, double* ms_total, int64_t* launches, mmk_stream_t stream);

/* > 0 when the plan runs all steps of a call inside one persistent kernel (1 csrc/wavenet_persist.hip, 2 wavenet_chain.hip,
 * 4 wavenet_lpipe.hip, 5 wavenet_spipe.hip, 6 wavenet_bpipe.hip; 3 was the XCD-pipelined kernel, removed in round 5: no geometry where it ran and won), 0 when it enqueues one fused kernel per layer half (hipGraph-replayed) */
int mmk_wavenet_mode(const mmk_wavenet_plan* plan);
/* 1 when the plan's last mode-5 launch streamed the clips through the stages two at a time (csrc/wavenet_spipe_pair.inc: an even number of clips,
 * 54 to 128 of them; the plan switch MMK_WN_SPIPE_PAIR=0 / 1 refuses / asks for it from 24 clips on), 0 otherwise */
int mmk_wavenet_pair_visits(const mmk_wavenet_plan* plan);
/* waits for `stream`; MMK_ERR_STATE if a hand-off inside the persistent kernel timed out */
int mmk_wavenet_sync_status(mmk_wavenet_plan* plan, mmk_stream_t stream);
/* Fault injection for the callers' tests: marks the plan as if a hand-off of its last call had timed out, so that the next
 * mmk_wavenet_sync_status fails exactly as it would after a real time-out (and the caller's redo path runs). */
int mmk_wavenet_inject_sync_error(mmk_wavenet_plan* plan, mmk_stream_t stream);

/* ------------------------------------------------------------------------
 * SampleRNN (mimikit/networks/sample_rnn_v2.py)
 * ---------------------------------------------------------------------- */
typedef struct mmk_srnn_config {
  int32_t n_tiers;                         /* len(frame_sizes) (last tier has no RNN) */
  int32_t frame_size[MMK_MAX_TIERS];
  int32_t hidden_dim;
  int32_t rnn_kind;                        /* 0 lstm, 1 gru, 2 rnn(tanh) */
  int32_t rnn_bias;
  int32_t h0_ones;                         /* h0_init == "ones" */
  int32_t q_levels;
  int32_t mlp_hidden, mlp_n_hidden, learn_temp;
  int32_t mlp_act;                         /* MLPIO.activation of every MLP head as an MMK_ACT_* code (Mish; anything else: the kernels in turns) */
  float min_temp;
  int32_t max_batch;
  int32_t n_rnn;                           /* Config.n_rnn: stacked recurrent layers per tier (:65, nn.LSTM / GRU num_layers); 0 = 1 */
  int32_t exec_mode;                       /* 0 = the library chooses (resident mode: the bottom tier's launch beside the tier kernels of a
                                            * second stream, where they are co-resident), 1 = the kernels in turns on one stream: what a
                                            * caller asks for to redo a batch after mmk_srnn_sync_status reported a timed-out wait */
  /* more than one input / target (from_config, sample_rnn_v2.py:141-145, :160-173, :181-182): every tier's input module is a
   * ZipReduceVariables (modules/io.py:289-313) over one framed linear per input, reduced with the weights of `inputs_mode`; the bottom
   * tier's hidden vector goes through one output module per target, and the loop writes output k into input k (loops/generate.py:213-218).
   * n_inputs / n_targets in [0, MMK_MAX_STREAMS], 0 = 1, n_targets <= n_inputs.  in_class[m]: classes of input m (0: q_levels).  Target 0 is
   * the head described above; target k >= 1 an MLPIO of x_q_levels[k] classes (entry 0 of the x_ arrays is unused).  Such networks run
   * with one launch per operation (mmk_srnn_warmup_multi / mmk_srnn_generate_multi). */
  int32_t n_inputs, n_targets;
  int32_t inputs_mode;                     /* ZipMode: 0 sum, 1 mean, 2 static_mix (softmax of the bound "tiers.i.input_module.weights") */
  int32_t in_class[MMK_MAX_STREAMS];
  int32_t x_q_levels[MMK_MAX_STREAMS], x_mlp_hidden[MMK_MAX_STREAMS], x_mlp_n_hidden[MMK_MAX_STREAMS], x_learn_temp[MMK_MAX_STREAMS];
  float x_min_temp[MMK_MAX_STREAMS];
  char tuning[MMK_TUNING_CHARS];           /* execution switches of THIS plan as "NAME=VALUE;NAME=VALUE" (empty: the library's choices), e.g.
                                            * "MMK_SRNN_RESIDENT=0" - what the parity tests use to put one network on every kernel that can run it.  The
                                            * library reads no environment variable (the diagnostic build, -DMMK_DIAG, falls back to it) */
} mmk_srnn_config;

typedef struct mmk_srnn_plan mmk_srnn_plan;

int mmk_srnn_plan_create(const mmk_srnn_config* cfg, mmk_srnn_plan** out);
void mmk_srnn_plan_destroy(mmk_srnn_plan* plan);
int mmk_srnn_plan_bind(mmk_srnn_plan* plan, const char* key, const float* dev_ptr, int64_t numel);
size_t mmk_srnn_workspace_bytes(const mmk_srnn_plan* plan);
int mmk_srnn_commit(mmk_srnn_plan* plan, void* workspace, size_t workspace_bytes, mmk_stream_t stream);
/* SampleRNN.reset_hidden (:266-268) */
int mmk_srnn_reset(mmk_srnn_plan* plan, mmk_stream_t stream);
/* SampleRNN.before_generate warm-up (:226-234): runs the tier schedule for
 * t in [rf, prompt_len - prompt_len % rf) on windows shifted by prompt_len % rf */
int mmk_srnn_warmup(mmk_srnn_plan* plan, int32_t batch, const int64_t* idx, int64_t idx_row_stride,
                    int64_t prompt_len, mmk_stream_t stream);
/* n_steps of the generate loop fused with SampleRNN.generate_step (:236-260) */
int mmk_srnn_generate(mmk_srnn_plan* plan, int32_t batch, int64_t* idx, int64_t idx_row_stride, int64_t t0,
                      int64_t n_steps, const float* temperature, const float* uniforms, mmk_stream_t stream);
int mmk_srnn_last_logits(mmk_srnn_plan* plan, int32_t batch, float* out, int64_t ld, mmk_stream_t stream);
/* the same calls for n_inputs class streams: idx[m] / idx_row_stride[m] for m < n_inputs; the class of target k is written IN PLACE to
 * idx[k][:, t]; uniforms: (n_targets, batch, n_steps).  With one input they are the calls above. */
int mmk_srnn_warmup_multi(mmk_srnn_plan* plan, int32_t batch, const int64_t* const* idx, const int64_t* idx_row_stride,
                          int64_t prompt_len, mmk_stream_t stream);
int mmk_srnn_generate_multi(mmk_srnn_plan* plan, int32_t batch, int64_t* const* idx, const int64_t* idx_row_stride, int64_t t0,
                            int64_t n_steps, const float* temperature, const float* uniforms, mmk_stream_t stream);
int mmk_srnn_last_logits_of(mmk_srnn_plan* plan, int32_t target, int32_t batch, float* out, int64_t ld, mmk_stream_t stream);
/* waits for the stream; fails (and clears the word) if a wait inside the tier / bottom kernels timed out since the last call -
 * the samples of that generation are invalid */
int mmk_srnn_sync_status(mmk_srnn_plan* plan, mmk_stream_t stream);
/* fault injection for the callers' tests: the next mmk_srnn_sync_status reports a timed-out wait (once) */
int mmk_srnn_inject_sync_error(mmk_srnn_plan* plan, mmk_stream_t stream);
/* diagnostic: generate blocks this plan has run in resident mode (the bottom tier as one launch beside the tier kernels of a
 * second stream) since it was created; tests assert that the mode they mean to cover is the one that ran */
int64_t mmk_srnn_resident_blocks(const mmk_srnn_plan* plan);
/* diagnostic: warm-ups (mmk_srnn_warmup: SampleRNN.before_generate, sample_rnn_v2.py:226-234) this plan has run as ONE teacher-forced resident launch - the
 * tiers with their matrices in registers, windows from the prompt, no bottom tier - instead of one launch per tier update */
int64_t mmk_srnn_resident_warmups(const mmk_srnn_plan* plan);

/* ------------------------------------------------------------------------
 * Seq2SeqLSTMNetwork (mimikit/networks/s2s_lstm_v2.py)
 * ---------------------------------------------------------------------- */
typedef struct mmk_s2s_config {
  int32_t in_dim;                          /* n_bins of the magspec input */
  int32_t out_dim;                         /* n_bins of the target */
  int32_t model_dim;
  int32_t hop;
  int32_t enc_n_lstm, dec_n_lstm;          /* bi-LSTM layers per side (1 .. 8) */
  int32_t out_abs;                         /* output activation Abs */
  int32_t max_batch;
  int32_t enc_downsampling;                /* 0 edge_sum, 1 edge_mean, 2 sum, 3 mean  (s2s_lstm_v2.py:105-113) */
  int32_t dec_upsampling;                  /* 0 linear_resample, 1 repeat             (:158-163) */
  int32_t enc_apply_residuals;             /* x = x + y from the second encoder layer on (:101-104) */
  int32_t dec_apply_residuals;             /* x = x + y after every decoder layer       (:175-178) */
  /* discrete IO (IOSpec.mulaw_io with an embedding input, tests/test_seq2seq.py:149-154): */
  int32_t in_classes;                      /* > 0: inputs are class indices through nn.Embedding(in_classes, model_dim)
                                              under ZipReduceVariables (:205-210); in_dim must equal model_dim */
  int32_t head_kind;                       /* 0: Linear [+ Abs] to out_dim bins; 1: MLP (networks/mlp.py:42-63) over out_dim
                                              classes, then the argmax of CategoricalSampler (modules/targets.py:43-44) */
  int32_t mlp_hidden, mlp_n_hidden;        /* head_kind 1: width, number of extra hidden blocks (0 .. 4) */
  int32_t mlp_act;                         /* head_kind 1: MLPIO.activation as an MMK_ACT_* code (Mish) */
  int32_t learn_temp;                      /* head_kind 1: one more output, logits / max(sigmoid(it), min_temp) */
  float min_temp;
  int32_t exec_mode;                       /* 0 = the library chooses (one resident launch per bi-LSTM layer where its workgroups are
                                            * co-resident), 1 = one launch per frame: what a caller asks for to redo a call after
                                            * mmk_s2s_sync_status reported a timed-out wait */
  char tuning[MMK_TUNING_CHARS];           /* execution switches of THIS plan as "NAME=VALUE;NAME=VALUE" (empty: the library's choices), e.g.
                                            * "MMK_S2S_SEQ=0" - what the parity tests use to put one network on every kernel that can run it.  The
                                            * library reads no environment variable (the diagnostic build, -DMMK_DIAG, falls back to it) */
} mmk_s2s_config;

typedef struct mmk_s2s_plan mmk_s2s_plan;

int mmk_s2s_plan_create(const mmk_s2s_config* cfg, mmk_s2s_plan** out);
void mmk_s2s_plan_destroy(mmk_s2s_plan* plan);
int mmk_s2s_plan_bind(mmk_s2s_plan* plan, const char* key, const float* dev_ptr, int64_t numel);
size_t mmk_s2s_workspace_bytes(const mmk_s2s_plan* plan);
int mmk_s2s_commit(mmk_s2s_plan* plan, void* workspace, size_t workspace_bytes, mmk_stream_t stream);
/* Seq2SeqLSTMNetwork.generate_step == forward (:246-266): x (batch, hop, in_dim)
 * -> y (batch, hop, out_dim).  Row strides are per frame, batch strides per clip. */
int mmk_s2s_step(mmk_s2s_plan* plan, int32_t batch, const float* x, int64_t x_batch_stride,
                 int64_t x_frame_stride, float* y, int64_t y_batch_stride, int64_t y_frame_stride,
                 mmk_stream_t stream);
/* n_calls successive generate_steps on one (batch, T, n_bins) tensor, in place:
 * call i reads frames [t0 + i*hop - hop, t0 + i*hop) and writes up to hop
 * frames at t0 + i*hop (clipped at t_total), as loops/generate.py:207-219 does. */
int mmk_s2s_generate(mmk_s2s_plan* plan, int32_t batch, float* frames, int64_t batch_stride,
                     int64_t frame_stride, int64_t t0, int64_t n_steps, int64_t t_total, mmk_stream_t stream);

/* The same two calls for a plan with in_classes > 0 and head_kind 1: x / y / classes hold int64 class indices, one per
 * (clip, position); generate_step hands the sampler no temperature (s2s_lstm_v2.py:262-263), so every class is an argmax. */
int mmk_s2s_step_classes(mmk_s2s_plan* plan, int32_t batch, const int64_t* x, int64_t x_batch_stride,
                         int64_t x_elem_stride, int64_t* y, int64_t y_batch_stride, int64_t y_elem_stride,
                         mmk_stream_t stream);
int mmk_s2s_generate_classes(mmk_s2s_plan* plan, int32_t batch, int64_t* classes, int64_t batch_stride,
                             int64_t elem_stride, int64_t t0, int64_t n_steps, int64_t t_total, mmk_stream_t stream);
/* the MLP head's outputs (before the learned-temperature division) of the last step: (batch * hop, out_dim + learn_temp)
 * rows (clip-major) copied to `out` with leading dimension out_ld - what the parity tests compare with the oracle's */
int mmk_s2s_last_logits(mmk_s2s_plan* plan, int32_t batch, float* out, int64_t out_ld, mmk_stream_t stream);
/* waits for the stream; fails (and clears the word) if a wait inside the resident bi-LSTM kernel (csrc/lstm_seq.hip) timed out
 * since the last call - the outputs of the calls in between are invalid */
int mmk_s2s_sync_status(mmk_s2s_plan* plan, mmk_stream_t stream);
/* fault injection for the callers' tests: the next mmk_s2s_sync_status reports a timed-out wait (once) */
int mmk_s2s_inject_sync_error(mmk_s2s_plan* plan, mmk_stream_t stream);
/* diagnostic: bi-LSTM layers this plan has run as ONE resident launch since it was created */
int64_t mmk_s2s_resident_launches(const mmk_s2s_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* MMK_H_ */
