// Python bindings (torch extension `kvmini._C`) of the gfx950 kernels.  Every function
// validates shapes/dtypes/devices on the host — an out-of-bounds access in a kernel can
// take the whole node down — and launches on PyTorch's current HIP stream, so calls are
// capturable into a HIP graph.

#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime.h>

extern "C" {
size_t kv_stream_state_bytes();
size_t kv_radix_state_bytes();
int kv_token_gaps(const long long* ev, int n, unsigned long long* keys, long long* vals, int slots, long long* last_t, long long n_last, float* gaps, hipStream_t stream);
int kv_stream_hist(const void* x, long long n, long long row_stride, int n_metrics, int dtype, unsigned int* hist, void* state, float lo_log2, float hi_log2,
                   hipStream_t stream);
int kv_radix_init(const void* stream_state, void* st, const double* ps, int n_ps, int n_metrics, hipStream_t stream);
int kv_radix_select(const void* x, long long n, long long row_stride, int n_metrics, int dtype, void* st, unsigned int* hist, hipStream_t stream);
int kv_radix_num_passes(int dtype);
int kv_radix_pass_hist(const void* x, long long n, long long row_stride, int n_metrics, int dtype, const void* st, unsigned int* hist, int pass, hipStream_t stream);
int kv_radix_pass_scan(void* st, unsigned int* hist, int n_metrics, int dtype, int pass, hipStream_t stream);
int kv_equal_hist(const void* x, long long n, long long row_stride, int n_metrics, int dtype, const void* state, unsigned long long* out, int n_bins,
                  hipStream_t stream);
int kv_energy_trapz(const double* ts, const double* watts, const int* n, int n_series, int max_samples, double t0, double t1, double* out, hipStream_t stream);
int kv_add_rmsnorm(const void* x, void* residual, const void* w, void* y, int T, int H, float eps, hipStream_t stream);
int kv_rope_kv_write(void* qkv, const int* positions, const float* cos_t, const float* sin_t, void* k_cache, void* v_cache, const int* slot_mapping, int T,
                     int nh, int nkv, int hd, int block_size, const float* kv_scale, hipStream_t stream);
int kv_paged_attention(const void* q, long long q_row_stride, const void* k_cache, const void* v_cache, const int* block_tables, int max_blocks_per_seq,
                       const int* tok_seq, const int* ctx_lens, void* out, int T, int nh, int nkv, int hd, int block_size, float scale, const float* kv_scale,
                       hipStream_t stream);
int kv_prefill_attention(const void* q, long long q_row_stride, const void* k_cache, const void* v_cache, const int* block_tables, int max_blocks_per_seq,
                         const int* tile_row0, const int* tile_n, const int* tile_pos0, const int* tile_seq, void* out, int n_tiles, int nh, int nkv, int hd,
                         int block_size, float scale, const float* kv_scale, int quads, hipStream_t stream);
int kv_prefill_attention_split(const void* q, long long q_row_stride, const void* k_cache, const void* v_cache, const int* block_tables, int max_blocks_per_seq,
                               const int* tile_row0, const int* tile_n, const int* tile_pos0, const int* tile_seq, void* out, int n_tiles, int nh, int nkv, int hd,
                               int block_size, float scale, const float* kv_scale, int n_split, float* part_o, float* part_ml, hipStream_t stream);
int kv_split_decode_attention(const void* q, long long q_row_stride, const void* k_cache, const void* v_cache, const int* block_tables, int max_blocks_per_seq,
                              const int* tok_seq, const int* ctx_lens, const int* row_list, int n_rows, void* out, int nh, int nkv, int hd, int block_size, float scale,
                              int waves, const float* kv_scale, int n_split, float* part_o, float* part_ml, int* counters, hipStream_t stream);
int kv_decode_attention_fused(const float* slabs, int n_slabs, long long slab_stride, const int* positions, const float* cos_t, const float* sin_t, void* k_cache,
                              void* v_cache, const int* slot_mapping, const int* block_tables, int max_blocks_per_seq, const int* tok_seq, const int* ctx_lens,
                              void* out, int T, int nh, int nkv, int hd, int block_size, const float* row_ss, int ss_parts, int h_in, float eps, float scale,
                              int waves, const float* kv_scale, int n_split, float* part_o, float* part_ml, int* counters, const void* pf_base,
                              const long long* pf, unsigned int* pf_sink, hipStream_t stream);
int kv_embed_fed(const void* table, const int* tokens, const int* feed, const long long* prev_ids, int n_prev, void* out, int T, int H, int V, hipStream_t stream);
int kv_argmax(const void* logits, long long row_stride, int S, int V, long long* out, hipStream_t stream);
int kv_sample_rows(const void* logits, long long row_stride, int V, const long long* rows, int R, const float* temperature, const int* top_k,
                   const float* top_p, const float* uniform, long long* out, hipStream_t stream);
int kv_guided_mask(void* logits, long long row_stride, int V, int n_cls, const long long* rows, const int* slots, const int* init, int R, int* state,
                   const unsigned char* cls, int vc, const int* script, int* cursor, const int* cur_init, int L, hipStream_t stream);
int kv_guided_advance(const long long* out_ids, const long long* rows, const int* slots, int R, int* state, const unsigned char* cls, int n_cls, const int* script,
                      int* cursor, int L, hipStream_t stream);
int kv_scale_rows(const void* x, const float* row_ss, int ss_parts, float eps, void* y, int T, int H, hipStream_t stream);
int kv_slab_add_bias(float* slab, long long row_stride, const float* bias, const float* row_ss, int ss_parts, float inv_h, float eps, int T, int N, hipStream_t stream);
int kv_swiglu_interleaved(const void* gu, void* act, int T, int I, hipStream_t stream);
int kv_pack_weight(const void* W, void* Wp, int N, int K, hipStream_t stream);
int kv_stream_gemm_plan(int N, int K, int ks_per_chunk, int max_splits, int* chunks_per_split);
int kv_stream_gemm(const void* Wp, const void* x, void* out, int M, int N, int K, long long ldx, long long ldy, int epi, int ks_per_chunk, int splits,
                   int chunks_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, hipStream_t stream);
int kvx_max_wg();
void* kvx_create(int world, int rank, long long slot_bytes);
int kvx_handle_bytes();
int kvx_get_handle(void* h, void* out);
int kvx_open_peer(void* h, int r, const void* handle);
int kvx_ready(void* h);
long long kvx_slot_bytes(void* h);
long long kvx_calls(void* h);
int kvx_parity(void* h);
int kvx_errors(void* h);
int kvx_allgather(void* h, const void* x, void* out, long long nbytes, hipStream_t stream);
int kvx_allreduce(void* h, const void* x, void* out, long long n, int two_shot, hipStream_t stream);
int kvx_residual_allreduce(void* h, const float* slabs, int n_slabs, long long slab_stride, const void* x, void* residual, int has_residual, float* ss_part, int T,
                           int H, hipStream_t stream);
int kvx_destroy(void* h);
void kv_set_attention_mfma(int on);
void kv_set_attention_mfma_unsplit(int on);
int kv_get_attention_mfma();
int kv_mfma_scale_probe(const int* a, const int* b, const int* sa, const int* sb, float* out, int fmt_a, int fmt_b, hipStream_t stream);
int kv_quantize_rows_fp8(const void* x, long long ldx, int M, int K, void* x8, float* scale, int permute, hipStream_t stream);
int kv_wide_gemm_w4a8(const void* Wp4, const int* w_blk, const void* x8, void* out, int M, int N, int K, long long ldx8, long long ldy, int epi, int bm, int bn, int variant,
                      int splits, int stages_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* a_scale, int m_fastest,
                      hipStream_t stream);
int kv_wide_gemm_fp8a(const void* Wp, const void* x8, void* out, int M, int N, int K, long long ldx8, long long ldy, int epi, int bm, int bn, int variant, int splits,
                      int stages_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, const float* a_scale,
                      int m_fastest, hipStream_t stream);
int kv_wide_gemm_stage_ksteps(int bm, int bn, int variant);
int kv_wide_gemm(const void* Wp, const void* x, void* out, int M, int N, int K, long long ldx, long long ldy, int epi, int bm, int bn, int variant, int splits,
                 int stages_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, int m_fastest,
                 hipStream_t stream);
int kv_stream_gemm3_plan(int N, int K, int nb, int nw, int max_splits, int target_wgs, int* groups_per_split);
int kv_stream_gemm3(const void* Wp, const void* x, void* out, int M, int N, int K, long long ldx, long long ldy, int epi, int nb, int nw, int splits,
                    int groups_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, int weight_bits, void* residual, int has_residual,
                    float* ss_out, int* counters, int ft, hipStream_t stream);
int kv_prefetch(const void* p, long long bytes, int workgroups, unsigned int* sink, hipStream_t stream);
int kv_pack_weight_fp8(const void* W, void* Wp, int N, int K, hipStream_t stream);
int kv_unpack_mxfp4(const void* Wp4, const void* scales, void* Wp, int N, int K, hipStream_t stream);
int kv_stream_gemm_rowsum(const void* Wp, const void* x, int M, int N, int K, long long ldx, int nb, void* residual, int has_residual, float* ss_out,
                          const float* w_scale, hipStream_t stream);
long long kv_decode_layers_workspace_bytes(int H, int nh, int nkv, int I, int s_qkv, int s_proj);
void kv_decode_layers_layout(int H, int nh, int nkv, int I, int s_qkv, int s_proj, long long* out9);
long long kv_decode_layers_state_offset(int H, int nh, int nkv, int I, int s_qkv, int s_proj);
int kv_decode_layers(const void* layers, int n_layers, int T, int H, int nh, int nkv, int I, const int* plans, const int* positions, const float* cos_t,
                     const float* sin_t, const int* slot_mapping, const int* block_tables, int max_blocks_per_seq, const int* tok_seq, const int* ctx_lens, float eps,
                     float scale, const float* kv_scale, void* residual, float* ss, void* ws, int grid, int timeline, hipStream_t stream);
int kv_residual_reduce(const float* slabs, int n_slabs, long long slab_stride, const void* x, void* residual, int has_residual, float* ss_part, int T, int H,
                       const void* pf_base, const long long* pf, unsigned int* pf_sink, hipStream_t stream);
int kv_add_rmsnorm_slabs(const float* slabs, int n_slabs, long long slab_stride, void* residual, const void* w, void* y, int T, int H, float eps,
                         hipStream_t stream);
int kv_rope_kv_write_slabs(const float* slabs, int n_slabs, long long slab_stride, void* q_out, const int* positions, const float* cos_t, const float* sin_t,
                           void* k_cache, void* v_cache, const int* slot_mapping, int T, int nh, int nkv, int hd, int block_size, const float* row_ss,
                           int ss_parts, int h_in, float eps, const float* kv_scale, hipStream_t stream);
}

namespace {

// Every wrapper evaluates cur_stream() as an argument of its launcher, i.e. right before the
// launch: clear the thread's sticky HIP error here so that the launcher's hipGetLastError()
// reports its own launch and not a stale error of an unrelated earlier runtime call.
hipStream_t cur_stream() {
  (void)hipGetLastError();
  return c10::hip::getCurrentHIPStream().stream();
}

void check(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed: ", rc < 0 ? "unsupported shape/arguments" : hipGetErrorString((hipError_t)rc));
}

#define CHECK_DEV(t) TORCH_CHECK((t).is_cuda(), #t " must be a GPU tensor")
#define CHECK_CONTIG(t) TORCH_CHECK((t).is_contiguous(), #t " must be contiguous")
#define CHECK_BF16(t) TORCH_CHECK((t).scalar_type() == at::kBFloat16, #t " must be bfloat16")
#define CHECK_I32(t) TORCH_CHECK((t).scalar_type() == at::kInt, #t " must be int32")

int sample_dtype(const at::Tensor& x) {
  if (x.scalar_type() == at::kFloat) return 0;
  if (x.scalar_type() == at::kDouble) return 1;
  TORCH_CHECK(false, "samples must be float32 or float64");
}

// ---- latency statistics ------------------------------------------------------------
int64_t stream_state_bytes() { return (int64_t)kv_stream_state_bytes(); }
int64_t radix_state_bytes() { return (int64_t)kv_radix_state_bytes(); }

// events int64 [n, 3] (request id, token index, t_ns) on the device -> gaps fp32 [n] in ms (NaN: no gap);
// last_t int64 [n_last] is the per-request time table that persists between calls; scratch_keys /
// scratch_vals int64 [slots], slots a power of two >= 2 n.
void token_gaps(at::Tensor events, at::Tensor scratch_keys, at::Tensor scratch_vals, at::Tensor last_t, at::Tensor gaps) {
  CHECK_DEV(events); CHECK_DEV(scratch_keys); CHECK_DEV(scratch_vals); CHECK_DEV(last_t); CHECK_DEV(gaps);
  CHECK_CONTIG(events); CHECK_CONTIG(scratch_keys); CHECK_CONTIG(scratch_vals); CHECK_CONTIG(last_t); CHECK_CONTIG(gaps);
  TORCH_CHECK(events.scalar_type() == at::kLong && events.dim() == 2 && events.size(1) == 3, "events must be int64 [n, 3]");
  TORCH_CHECK(scratch_keys.scalar_type() == at::kLong && scratch_vals.scalar_type() == at::kLong && last_t.scalar_type() == at::kLong, "tables must be int64");
  TORCH_CHECK(gaps.scalar_type() == at::kFloat && gaps.numel() >= events.size(0), "gaps must be fp32 [n]");
  const int64_t n = events.size(0), slots = scratch_keys.numel();
  TORCH_CHECK(n < (1 << 30) && scratch_vals.numel() == slots, "scratch tables of equal size");
  if (n == 0) return;
  check((int)hipMemsetAsync(scratch_keys.data_ptr(), 0, (size_t)slots * 8, cur_stream()), "hipMemsetAsync");
  check(kv_token_gaps((const long long*)events.data_ptr(), (int)n, (unsigned long long*)scratch_keys.data_ptr(), (long long*)scratch_vals.data_ptr(), (int)slots,
                      (long long*)last_t.data_ptr(), last_t.numel(), (float*)gaps.data_ptr(), cur_stream()),
        "kv_token_gaps");
}

void stream_hist(at::Tensor x, int64_t n, at::Tensor hist, at::Tensor state, double lo_log2, double hi_log2) {
  CHECK_DEV(x); CHECK_DEV(hist); CHECK_DEV(state); CHECK_CONTIG(x); CHECK_CONTIG(hist); CHECK_CONTIG(state);
  TORCH_CHECK(x.dim() == 2, "x must be [n_metrics, row_stride]");
  const int64_t m = x.size(0);
  TORCH_CHECK(n >= 0 && n <= x.size(1), "n exceeds the row length");
  TORCH_CHECK(hist.scalar_type() == at::kInt && hist.numel() == m * 2048, "hist must be int32 [n_metrics, 2048]");
  TORCH_CHECK(state.scalar_type() == at::kByte && state.numel() == m * stream_state_bytes(), "state must be uint8 [n_metrics * state_bytes]");
  TORCH_CHECK(hi_log2 > lo_log2, "empty bin range");
  check(kv_stream_hist(x.data_ptr(), n, x.size(1), (int)m, sample_dtype(x), (unsigned int*)hist.data_ptr(), state.data_ptr(), (float)lo_log2, (float)hi_log2,
                       cur_stream()),
        "kv_stream_hist");
}

void radix_select(at::Tensor x, int64_t n, at::Tensor stream_state, at::Tensor ps, at::Tensor rstate, at::Tensor hist) {
  CHECK_DEV(x); CHECK_DEV(stream_state); CHECK_DEV(ps); CHECK_DEV(rstate); CHECK_DEV(hist);
  CHECK_CONTIG(x); CHECK_CONTIG(stream_state); CHECK_CONTIG(ps); CHECK_CONTIG(rstate); CHECK_CONTIG(hist);
  TORCH_CHECK(x.dim() == 2, "x must be [n_metrics, row_stride]");
  const int64_t m = x.size(0);
  TORCH_CHECK(n >= 0 && n <= x.size(1), "n exceeds the row length");
  TORCH_CHECK(ps.scalar_type() == at::kDouble && ps.numel() >= 1 && ps.numel() <= 4, "ps must be float64 with 1..4 quantile levels");
  TORCH_CHECK(stream_state.scalar_type() == at::kByte && stream_state.numel() == m * stream_state_bytes(), "bad stream state size");
  TORCH_CHECK(rstate.scalar_type() == at::kByte && rstate.numel() == m * radix_state_bytes(), "bad radix state size");
  TORCH_CHECK(hist.scalar_type() == at::kInt && hist.numel() == m * 8 * 2048, "hist must be int32 [n_metrics, 8, 2048] (zeroed)");
  auto s = cur_stream();
  check(kv_radix_init(stream_state.data_ptr(), rstate.data_ptr(), (const double*)ps.data_ptr(), (int)ps.numel(), (int)m, s), "kv_radix_init");
  check(kv_radix_select(x.data_ptr(), n, x.size(1), (int)m, sample_dtype(x), rstate.data_ptr(), (unsigned int*)hist.data_ptr(), s), "kv_radix_select");
}

// ---- the selection one pass at a time (samples spread over ranks: the digit histograms are summed between hist and scan)
int64_t radix_num_passes(bool is_double) { return kv_radix_num_passes(is_double ? 1 : 0); }

void radix_init(at::Tensor stream_state, at::Tensor ps, at::Tensor rstate) {
  CHECK_DEV(stream_state); CHECK_DEV(ps); CHECK_DEV(rstate); CHECK_CONTIG(stream_state); CHECK_CONTIG(ps); CHECK_CONTIG(rstate);
  TORCH_CHECK(ps.scalar_type() == at::kDouble && ps.numel() >= 1 && ps.numel() <= 4, "ps must be float64 with 1..4 quantile levels");
  const int64_t m = stream_state.numel() / stream_state_bytes();
  TORCH_CHECK(stream_state.scalar_type() == at::kByte && m >= 1 && stream_state.numel() == m * stream_state_bytes(), "bad stream state size");
  TORCH_CHECK(rstate.scalar_type() == at::kByte && rstate.numel() == m * radix_state_bytes(), "bad radix state size");
  check(kv_radix_init(stream_state.data_ptr(), rstate.data_ptr(), (const double*)ps.data_ptr(), (int)ps.numel(), (int)m, cur_stream()), "kv_radix_init");
}

void radix_pass_hist(at::Tensor x, int64_t n, at::Tensor rstate, at::Tensor hist, int64_t pass) {
  CHECK_DEV(x); CHECK_DEV(rstate); CHECK_DEV(hist); CHECK_CONTIG(x); CHECK_CONTIG(rstate); CHECK_CONTIG(hist);
  TORCH_CHECK(x.dim() == 2, "x must be [n_metrics, row_stride]");
  const int64_t m = x.size(0);
  TORCH_CHECK(n >= 0 && n <= x.size(1), "n exceeds the row length");
  TORCH_CHECK(rstate.scalar_type() == at::kByte && rstate.numel() == m * radix_state_bytes(), "bad radix state size");
  TORCH_CHECK(hist.scalar_type() == at::kInt && hist.numel() == m * 8 * 2048, "hist must be int32 [n_metrics, 8, 2048]");
  check(kv_radix_pass_hist(x.data_ptr(), n, x.size(1), (int)m, sample_dtype(x), rstate.data_ptr(), (unsigned int*)hist.data_ptr(), (int)pass, cur_stream()), "kv_radix_pass_hist");
}

void radix_pass_scan(at::Tensor rstate, at::Tensor hist, bool is_double, int64_t pass) {
  CHECK_DEV(rstate); CHECK_DEV(hist); CHECK_CONTIG(rstate); CHECK_CONTIG(hist);
  const int64_t m = rstate.numel() / radix_state_bytes();
  TORCH_CHECK(rstate.scalar_type() == at::kByte && m >= 1 && rstate.numel() == m * radix_state_bytes(), "bad radix state size");
  TORCH_CHECK(hist.scalar_type() == at::kInt && hist.numel() == m * 8 * 2048, "hist must be int32 [n_metrics, 8, 2048]");
  check(kv_radix_pass_scan(rstate.data_ptr(), (unsigned int*)hist.data_ptr(), (int)m, is_double ? 1 : 0, (int)pass, cur_stream()), "kv_radix_pass_scan");
}

void equal_hist(at::Tensor x, int64_t n, at::Tensor stream_state, at::Tensor out, int64_t n_bins) {
  CHECK_DEV(x); CHECK_DEV(stream_state); CHECK_DEV(out); CHECK_CONTIG(x); CHECK_CONTIG(stream_state); CHECK_CONTIG(out);
  TORCH_CHECK(x.dim() == 2, "x must be [n_metrics, row_stride]");
  const int64_t m = x.size(0);
  TORCH_CHECK(n >= 0 && n <= x.size(1), "n exceeds the row length");
  TORCH_CHECK(n_bins >= 1 && n_bins <= 64, "n_bins must be in 1..64");
  TORCH_CHECK(out.scalar_type() == at::kLong && out.numel() == m * 64, "out must be int64 [n_metrics, 64] (zeroed)");
  TORCH_CHECK(stream_state.scalar_type() == at::kByte && stream_state.numel() == m * stream_state_bytes(), "bad stream state size");
  check(kv_equal_hist(x.data_ptr(), n, x.size(1), (int)m, sample_dtype(x), stream_state.data_ptr(), (unsigned long long*)out.data_ptr(), (int)n_bins, cur_stream()),
        "kv_equal_hist");
}

at::Tensor energy_trapz(at::Tensor ts, at::Tensor watts, at::Tensor n, double t0, double t1) {
  CHECK_DEV(ts); CHECK_DEV(watts); CHECK_DEV(n); CHECK_CONTIG(ts); CHECK_CONTIG(watts); CHECK_CONTIG(n); CHECK_I32(n);
  TORCH_CHECK(ts.scalar_type() == at::kDouble && watts.scalar_type() == at::kDouble, "ts/watts must be float64");
  TORCH_CHECK(ts.dim() == 2 && watts.sizes() == ts.sizes() && n.numel() == ts.size(0), "shapes: ts/watts [series, samples], n [series]");
  TORCH_CHECK(n.max().item<int>() <= ts.size(1) && n.min().item<int>() >= 0, "n out of range");
  auto out = at::zeros({ts.size(0), 3}, ts.options());
  check(kv_energy_trapz((const double*)ts.data_ptr(), (const double*)watts.data_ptr(), (const int*)n.data_ptr(), (int)ts.size(0), (int)ts.size(1), t0, t1,
                        (double*)out.data_ptr(), cur_stream()),
        "kv_energy_trapz");
  return out;
}

// ---- engine ops ----------------------------------------------------------------------
// The paged KV cache is bf16, or — when kv_scale (fp32 [2, nkv]: K scales, then V scales) is
// given — one e4m3 byte per element holding x / scale.  Returns the scale pointer (null: bf16).
const float* check_cache(const at::Tensor& k_cache, const at::Tensor& v_cache, const c10::optional<at::Tensor>& kv_scale) {
  CHECK_DEV(k_cache); CHECK_DEV(v_cache); CHECK_CONTIG(k_cache); CHECK_CONTIG(v_cache);
  TORCH_CHECK(k_cache.dim() == 4 && v_cache.sizes() == k_cache.sizes(), "cache must be [blocks, nkv, block_size, hd]");
  if (!kv_scale.has_value()) {
    CHECK_BF16(k_cache); CHECK_BF16(v_cache);
    return nullptr;
  }
  const at::Tensor& sc = *kv_scale;
  TORCH_CHECK(k_cache.element_size() == 1 && v_cache.element_size() == 1, "with kv_scale the cache holds one byte per element (float8_e4m3fn / uint8)");
  CHECK_DEV(sc); CHECK_CONTIG(sc);
  TORCH_CHECK(sc.scalar_type() == at::kFloat && sc.numel() == 2 * k_cache.size(1), "kv_scale must be fp32 [2, nkv]");
  return (const float*)sc.data_ptr();
}

void add_rmsnorm(at::Tensor x, at::Tensor residual, at::Tensor w, at::Tensor y, double eps) {
  CHECK_DEV(x); CHECK_BF16(x); CHECK_BF16(residual); CHECK_BF16(w); CHECK_BF16(y);
  CHECK_CONTIG(x); CHECK_CONTIG(residual); CHECK_CONTIG(w); CHECK_CONTIG(y);
  TORCH_CHECK(x.dim() == 2 && residual.sizes() == x.sizes() && y.sizes() == x.sizes() && w.numel() == x.size(1), "add_rmsnorm shape mismatch");
  TORCH_CHECK(x.size(1) % 8 == 0 && x.size(1) <= 32768, "hidden size must be a multiple of 8 and <= 32768");
  check(kv_add_rmsnorm(x.data_ptr(), residual.data_ptr(), w.data_ptr(), y.data_ptr(), (int)x.size(0), (int)x.size(1), (float)eps, cur_stream()), "kv_add_rmsnorm");
}

void rope_kv_write(at::Tensor qkv, at::Tensor positions, at::Tensor cos_t, at::Tensor sin_t, at::Tensor k_cache, at::Tensor v_cache, at::Tensor slot_mapping,
                   int64_t nh, int64_t nkv, int64_t hd, c10::optional<at::Tensor> kv_scale) {
  CHECK_DEV(qkv); CHECK_BF16(qkv); CHECK_I32(positions); CHECK_I32(slot_mapping);
  const float* ksc = check_cache(k_cache, v_cache, kv_scale);
  CHECK_CONTIG(qkv); CHECK_CONTIG(cos_t); CHECK_CONTIG(sin_t); CHECK_CONTIG(positions); CHECK_CONTIG(slot_mapping);
  const int64_t T = qkv.size(0);
  TORCH_CHECK(qkv.dim() == 2 && qkv.size(1) == (nh + 2 * nkv) * hd, "qkv must be [T, (nh + 2 nkv) * hd]");
  TORCH_CHECK(positions.numel() >= T && slot_mapping.numel() >= T, "positions/slot_mapping shorter than T");
  TORCH_CHECK(cos_t.scalar_type() == at::kFloat && sin_t.scalar_type() == at::kFloat && cos_t.dim() == 2 && cos_t.size(1) == hd / 2 && sin_t.sizes() == cos_t.sizes(),
              "cos/sin must be float32 [max_pos, hd/2]");
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(1) == nkv && k_cache.size(3) == hd && v_cache.sizes() == k_cache.sizes(), "cache must be [blocks, nkv, block_size, hd]");
  // positions and slots are device data: bounds are guaranteed by the scheduler (positions < max_model_len = rows of cos/sin; slots < blocks * block_size)
  check(kv_rope_kv_write(qkv.data_ptr(), (const int*)positions.data_ptr(), (const float*)cos_t.data_ptr(), (const float*)sin_t.data_ptr(), k_cache.data_ptr(),
                         v_cache.data_ptr(), (const int*)slot_mapping.data_ptr(), (int)T, (int)nh, (int)nkv, (int)hd, (int)k_cache.size(2), ksc, cur_stream()),
        "kv_rope_kv_write");
}

void paged_attention(at::Tensor q, at::Tensor k_cache, at::Tensor v_cache, at::Tensor block_tables, at::Tensor tok_seq, at::Tensor ctx_lens, at::Tensor out,
                     double scale, c10::optional<at::Tensor> kv_scale) {
  CHECK_DEV(q); CHECK_BF16(q); CHECK_BF16(out); CHECK_I32(block_tables); CHECK_I32(tok_seq); CHECK_I32(ctx_lens);
  const float* ksc = check_cache(k_cache, v_cache, kv_scale);
  CHECK_CONTIG(block_tables); CHECK_CONTIG(out); CHECK_CONTIG(tok_seq); CHECK_CONTIG(ctx_lens);
  TORCH_CHECK(q.dim() == 3 && q.stride(2) == 1 && q.stride(1) == q.size(2), "q must be [T, nh, hd] with contiguous heads (row stride free)");
  const int64_t T = q.size(0), nh = q.size(1), hd = q.size(2);
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(3) == hd && v_cache.sizes() == k_cache.sizes(), "cache must be [blocks, nkv, block_size, hd]");
  const int64_t nkv = k_cache.size(1);
  TORCH_CHECK(block_tables.dim() == 2 && out.dim() == 2 && out.size(0) == T && out.size(1) == nh * hd, "bad block_tables/out shape");
  TORCH_CHECK(tok_seq.numel() >= T && ctx_lens.numel() >= T, "tok_seq/ctx_lens shorter than T");
  check(kv_paged_attention(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), (const int*)block_tables.data_ptr(), (int)block_tables.size(1),
                           (const int*)tok_seq.data_ptr(), (const int*)ctx_lens.data_ptr(), out.data_ptr(), (int)T, (int)nh, (int)nkv, (int)hd,
                           (int)k_cache.size(2), (float)scale, ksc, cur_stream()),
        "kv_paged_attention");
}

// Tiled MFMA attention for prompt chunks; `tiles` is int32 [4, n_tiles]: row0, n, pos0, block-table row
// quads: the tile list is padded per chunk to a multiple of four tiles, so every aligned group of
// four is consecutive tiles of one chunk (unused ones last): the 8-wave kernel takes a group each.
void prefill_attention(at::Tensor q, at::Tensor k_cache, at::Tensor v_cache, at::Tensor block_tables, at::Tensor tiles, at::Tensor out, double scale,
                       c10::optional<at::Tensor> kv_scale, bool quads) {
  CHECK_DEV(q); CHECK_BF16(q); CHECK_BF16(out); CHECK_I32(block_tables); CHECK_I32(tiles);
  const float* ksc = check_cache(k_cache, v_cache, kv_scale);
  CHECK_CONTIG(block_tables); CHECK_CONTIG(out); CHECK_CONTIG(tiles);
  TORCH_CHECK(q.dim() == 3 && q.stride(2) == 1 && q.stride(1) == q.size(2) && q.stride(0) % 8 == 0, "q must be [T, nh, hd] with contiguous heads and 16-byte aligned rows");
  const int64_t T = q.size(0), nh = q.size(1), hd = q.size(2);
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(3) == hd && v_cache.sizes() == k_cache.sizes(), "cache must be [blocks, nkv, block_size, hd]");
  TORCH_CHECK(tiles.dim() == 2 && tiles.size(0) == 4, "tiles must be int32 [4, n_tiles]");
  TORCH_CHECK(block_tables.dim() == 2 && out.dim() == 2 && out.size(0) == T && out.size(1) == nh * hd, "bad block_tables/out shape");
  const int64_t n_tiles = tiles.size(1);
  TORCH_CHECK(!quads || (n_tiles % 4 == 0 && nh / k_cache.size(1) <= 4), "quads: a multiple of four tiles and head groups of at most 4");
  const int* tp = (const int*)tiles.data_ptr();
  check(kv_prefill_attention(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), (const int*)block_tables.data_ptr(), (int)block_tables.size(1), tp,
                             tp + n_tiles, tp + 2 * n_tiles, tp + 3 * n_tiles, out.data_ptr(), (int)n_tiles, (int)nh, (int)k_cache.size(1), (int)hd,
                             (int)k_cache.size(2), (float)scale, ksc, quads ? 1 : 0, cur_stream()),
        "kv_prefill_attention");
}

// The first `n_tiles` tiles of `tiles` ([4, max_tiles]) with every tile's key range in `n_split` shares; part_o / part_ml: fp32 workspaces
// of at least n_split * 16 * n_tiles * nh * 128 and ... * 2 elements.
void prefill_attention_split(at::Tensor q, at::Tensor k_cache, at::Tensor v_cache, at::Tensor block_tables, at::Tensor tiles, int64_t n_tiles, at::Tensor out, double scale,
                             c10::optional<at::Tensor> kv_scale, int64_t n_split, at::Tensor part_o, at::Tensor part_ml) {
  CHECK_DEV(q); CHECK_BF16(q); CHECK_BF16(out); CHECK_I32(block_tables); CHECK_I32(tiles); CHECK_DEV(part_o); CHECK_DEV(part_ml);
  const float* ksc = check_cache(k_cache, v_cache, kv_scale);
  CHECK_CONTIG(block_tables); CHECK_CONTIG(out); CHECK_CONTIG(tiles); CHECK_CONTIG(part_o); CHECK_CONTIG(part_ml);
  TORCH_CHECK(q.dim() == 3 && q.stride(2) == 1 && q.stride(1) == q.size(2) && q.stride(0) % 8 == 0, "q must be [T, nh, hd] with contiguous heads and 16-byte aligned rows");
  const int64_t T = q.size(0), nh = q.size(1), hd = q.size(2);
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(3) == hd && v_cache.sizes() == k_cache.sizes(), "cache must be [blocks, nkv, block_size, hd]");
  TORCH_CHECK(tiles.dim() == 2 && tiles.size(0) == 4 && n_tiles >= 0 && n_tiles <= tiles.size(1), "tiles must be int32 [4, max_tiles] with n_tiles <= max_tiles");
  TORCH_CHECK(block_tables.dim() == 2 && out.dim() == 2 && out.size(0) == T && out.size(1) == nh * hd, "bad block_tables/out shape");
  TORCH_CHECK(n_split >= 1 && n_split <= 64, "n_split must be 1..64");
  TORCH_CHECK(part_o.scalar_type() == at::kFloat && part_ml.scalar_type() == at::kFloat, "workspaces must be fp32");
  const int64_t rows = n_split * 16 * n_tiles;
  TORCH_CHECK(part_o.numel() >= rows * nh * hd && part_ml.numel() >= rows * nh * 2, "workspace too small for n_split x tiles");
  TORCH_CHECK(((uintptr_t)part_o.data_ptr() & 15) == 0 && ((uintptr_t)part_ml.data_ptr() & 7) == 0, "workspaces must be 16-byte aligned");
  const int* tp = (const int*)tiles.data_ptr();
  const int64_t mt = tiles.size(1);
  check(kv_prefill_attention_split(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), (const int*)block_tables.data_ptr(), (int)block_tables.size(1), tp,
                                   tp + mt, tp + 2 * mt, tp + 3 * mt, out.data_ptr(), (int)n_tiles, (int)nh, (int)k_cache.size(1), (int)hd, (int)k_cache.size(2),
                                   (float)scale, ksc, (int)n_split, (float*)part_o.data_ptr(), (float*)part_ml.data_ptr(), cur_stream()),
        "kv_prefill_attention_split");
}

const float* ss_ptr(const c10::optional<at::Tensor>& row_ss, int64_t rows, int* parts);

// Optional weight prefetch beside a latency-bound kernel (PrefetchPlan in engine_kernels.hip):
// `w` = the packed weights of the GEMM that runs next, `plan` = {gx, gy, nw, kib, tile_bytes,
// split_bytes, prefetch workgroups}, `sink` = an int32 scratch word.  Fills v[8] (the plan plus
// the byte size of `w`, against which the kernel launcher checks every range) and returns w's
// address, or nullptr when no prefetch was asked for.
const void* prefetch_args(const c10::optional<at::Tensor>& w, const c10::optional<std::vector<int64_t>>& plan, const c10::optional<at::Tensor>& sink, long long* v,
                          unsigned int** sink_ptr) {
  *sink_ptr = nullptr;
  if (!w.has_value() || !plan.has_value()) return nullptr;
  TORCH_CHECK(plan->size() == 7, "prefetch plan must have 7 entries");
  TORCH_CHECK(sink.has_value(), "prefetch needs a sink tensor");
  CHECK_DEV((*w)); CHECK_CONTIG((*w)); CHECK_DEV((*sink));
  TORCH_CHECK(sink->scalar_type() == at::kInt && sink->numel() >= 1, "sink must be an int32 tensor");
  for (int i = 0; i < 7; ++i) v[i] = (*plan)[i];
  v[7] = (long long)w->numel() * w->element_size();
  *sink_ptr = (unsigned int*)sink->data_ptr();
  return w->data_ptr();
}

// RoPE + KV write + paged attention in one launch (decode-only steps; see engine_kernels.hip)
void decode_attention_fused(at::Tensor slabs, at::Tensor positions, at::Tensor cos_t, at::Tensor sin_t, at::Tensor k_cache, at::Tensor v_cache,
                            at::Tensor slot_mapping, at::Tensor block_tables, at::Tensor tok_seq, at::Tensor ctx_lens, at::Tensor out, int64_t nh,
                            c10::optional<at::Tensor> row_ss, int64_t h_in, double eps, double scale, int64_t waves, c10::optional<at::Tensor> kv_scale,
                            int64_t n_split, c10::optional<at::Tensor> split_ws, c10::optional<at::Tensor> pf_w, c10::optional<std::vector<int64_t>> pf_plan,
                            c10::optional<at::Tensor> pf_sink) {
  TORCH_CHECK(waves == 4 || waves == 8, "waves per workgroup must be 4 or 8");
  long long pfv[8];
  unsigned int* pfs = nullptr;
  const void* pfb = prefetch_args(pf_w, pf_plan, pf_sink, pfv, &pfs);
  CHECK_DEV(slabs); CHECK_CONTIG(slabs); CHECK_I32(positions); CHECK_I32(slot_mapping); CHECK_I32(block_tables); CHECK_I32(tok_seq); CHECK_I32(ctx_lens);
  const float* ksc = check_cache(k_cache, v_cache, kv_scale);
  CHECK_BF16(out); CHECK_CONTIG(out); CHECK_CONTIG(block_tables);
  CHECK_CONTIG(positions); CHECK_CONTIG(slot_mapping); CHECK_CONTIG(tok_seq); CHECK_CONTIG(ctx_lens); CHECK_CONTIG(cos_t); CHECK_CONTIG(sin_t);
  TORCH_CHECK(slabs.scalar_type() == at::kFloat && slabs.dim() == 3, "slabs must be fp32 [S, T, W]");
  TORCH_CHECK(cos_t.scalar_type() == at::kFloat && sin_t.scalar_type() == at::kFloat, "cos/sin tables must be fp32");
  const int64_t T = slabs.size(1), W = slabs.size(2);
  TORCH_CHECK(k_cache.dim() == 4 && v_cache.sizes() == k_cache.sizes(), "cache must be [blocks, nkv, block_size, hd]");
  const int64_t nkv = k_cache.size(1), hd = k_cache.size(3);
  TORCH_CHECK(W == (nh + 2 * nkv) * hd, "slab width must be (nh + 2 nkv) * head_dim");
  TORCH_CHECK(cos_t.dim() == 2 && cos_t.size(1) == hd / 2 && sin_t.sizes() == cos_t.sizes(), "cos/sin tables must be [max_pos, hd/2]");
  TORCH_CHECK(positions.numel() >= T && slot_mapping.numel() >= T && tok_seq.numel() >= T && ctx_lens.numel() >= T, "per-token index tensors shorter than T");
  TORCH_CHECK(block_tables.dim() == 2 && out.dim() == 2 && out.size(0) == T && out.size(1) == nh * hd, "bad block_tables/out shape");
  int parts = 0;
  const float* ssp = ss_ptr(row_ss, T, &parts);
  // split-KV workspace, one fp32 tensor: 4096 int32 arrival counters first (zero at rest: the kernel resets
  // them; at a fixed place whatever T is), then (max, sum) pairs [T, nkv, n_split, rep, 2], then partial
  // outputs [T, nkv, n_split, rep*hd]
  float *part_o = nullptr, *part_ml = nullptr;
  int* counters = nullptr;
  TORCH_CHECK(n_split >= 1 && n_split <= 64, "n_split must be in [1, 64]");
  if (n_split > 1) {
    TORCH_CHECK(split_ws.has_value(), "split_ws is required with n_split > 1");
    const at::Tensor& ws = *split_ws;
    CHECK_DEV(ws); CHECK_CONTIG(ws);
    const int64_t rep = nh / nkv, n_o = T * nkv * n_split * rep * hd, n_ml = T * nkv * n_split * rep * 2;
    TORCH_CHECK(T * nkv <= 4096, "split-KV decode attention: at most 4096 (token, kv head) pairs");
    TORCH_CHECK(ws.scalar_type() == at::kFloat && ws.numel() >= 4096 + n_o + n_ml, "split_ws too small");
    counters = (int*)ws.data_ptr();
    part_ml = (float*)ws.data_ptr() + 4096;
    part_o = part_ml + n_ml;
  }
  check(kv_decode_attention_fused((const float*)slabs.data_ptr(), (int)slabs.size(0), T * W, (const int*)positions.data_ptr(), (const float*)cos_t.data_ptr(),
                                  (const float*)sin_t.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), (const int*)slot_mapping.data_ptr(),
                                  (const int*)block_tables.data_ptr(), (int)block_tables.size(1), (const int*)tok_seq.data_ptr(), (const int*)ctx_lens.data_ptr(),
                                  out.data_ptr(), (int)T, (int)nh, (int)nkv, (int)hd, (int)k_cache.size(2), ssp, parts, (int)h_in, (float)eps, (float)scale,
                                  (int)waves, ksc, (int)n_split, part_o, part_ml, counters, pfb, pfb ? pfv : nullptr, pfs, cur_stream()),
        "kv_decode_attention_fused");
}

// Split-KV attention for listed decode rows of a step that also carries prompt chunks (q rotated,
// keys / values of the step already in the cache).  `rows`: int32 token rows, -1 = unused entry.
void split_decode_attention(at::Tensor q, at::Tensor k_cache, at::Tensor v_cache, at::Tensor block_tables, at::Tensor tok_seq, at::Tensor ctx_lens, at::Tensor rows,
                            at::Tensor out, double scale, int64_t waves, c10::optional<at::Tensor> kv_scale, int64_t n_split, at::Tensor split_ws) {
  TORCH_CHECK(waves == 4 || waves == 8, "waves per workgroup must be 4 or 8");
  CHECK_DEV(q); CHECK_BF16(q); CHECK_BF16(out); CHECK_CONTIG(out); CHECK_I32(block_tables); CHECK_CONTIG(block_tables);
  CHECK_I32(tok_seq); CHECK_CONTIG(tok_seq); CHECK_I32(ctx_lens); CHECK_CONTIG(ctx_lens); CHECK_I32(rows); CHECK_CONTIG(rows);
  const float* ksc = check_cache(k_cache, v_cache, kv_scale);
  TORCH_CHECK(q.dim() == 3 && q.stride(2) == 1 && q.stride(1) == q.size(2) && q.stride(0) % 8 == 0, "q must be [T, nh, hd] with contiguous heads and 16-byte aligned rows");
  const int64_t T = q.size(0), nh = q.size(1), hd = q.size(2);
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(3) == hd && v_cache.sizes() == k_cache.sizes(), "cache must be [blocks, nkv, block_size, hd]");
  const int64_t nkv = k_cache.size(1), n_rows = rows.numel();
  TORCH_CHECK(block_tables.dim() == 2 && out.dim() == 2 && out.size(0) == T && out.size(1) == nh * hd, "bad block_tables/out shape");
  TORCH_CHECK(tok_seq.numel() >= T && ctx_lens.numel() >= T, "per-token index tensors shorter than T");
  TORCH_CHECK(n_split >= 1 && n_split <= 64, "n_split must be in [1, 64]");
  TORCH_CHECK(n_rows * nkv <= 4096, "split-KV decode attention: at most 4096 (row, kv head) pairs");
  CHECK_DEV(split_ws); CHECK_CONTIG(split_ws);
  const int64_t rep = nh / nkv, n_o = n_rows * nkv * n_split * rep * hd, n_ml = n_rows * nkv * n_split * rep * 2;
  TORCH_CHECK(split_ws.scalar_type() == at::kFloat && split_ws.numel() >= 4096 + n_o + n_ml, "split_ws too small");
  int* counters = (int*)split_ws.data_ptr();
  float* part_ml = (float*)split_ws.data_ptr() + 4096;
  float* part_o = part_ml + n_ml;
  check(kv_split_decode_attention(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), (const int*)block_tables.data_ptr(), (int)block_tables.size(1),
                                  (const int*)tok_seq.data_ptr(), (const int*)ctx_lens.data_ptr(), (const int*)rows.data_ptr(), (int)n_rows, out.data_ptr(), (int)nh,
                                  (int)nkv, (int)hd, (int)k_cache.size(2), (float)scale, (int)waves, ksc, (int)n_split, part_o, part_ml, counters, cur_stream()),
        "kv_split_decode_attention");
}

void embed_fed(at::Tensor table, at::Tensor tokens, at::Tensor feed, at::Tensor prev_ids, at::Tensor out) {
  CHECK_DEV(table); CHECK_BF16(table); CHECK_CONTIG(table); CHECK_DEV(out); CHECK_BF16(out); CHECK_CONTIG(out);
  CHECK_DEV(tokens); CHECK_DEV(feed); CHECK_DEV(prev_ids); CHECK_CONTIG(tokens); CHECK_CONTIG(feed); CHECK_CONTIG(prev_ids);
  TORCH_CHECK(table.dim() == 2 && out.dim() == 2 && out.size(1) == table.size(1) && table.size(1) % 8 == 0, "table [V, H], out [T, H], H % 8 == 0");
  const int64_t T = out.size(0);
  TORCH_CHECK(tokens.scalar_type() == at::kInt && feed.scalar_type() == at::kInt && tokens.numel() >= T && feed.numel() >= T, "tokens/feed must be int32 with >= T elements");
  TORCH_CHECK(prev_ids.scalar_type() == at::kLong && prev_ids.numel() >= 1, "prev_ids must be a non-empty int64 tensor");
  check(kv_embed_fed(table.data_ptr(), (const int*)tokens.data_ptr(), (const int*)feed.data_ptr(), (const long long*)prev_ids.data_ptr(), (int)prev_ids.numel(), out.data_ptr(), (int)T,
                     (int)table.size(1), (int)table.size(0), cur_stream()),
        "kv_embed_fed");
}

void argmax(at::Tensor logits, at::Tensor out) {
  CHECK_DEV(logits); CHECK_BF16(logits); CHECK_DEV(out); CHECK_CONTIG(out);
  TORCH_CHECK(logits.dim() == 2 && logits.stride(1) == 1, "logits must be [S, V] with unit inner stride");
  TORCH_CHECK(out.scalar_type() == at::kLong && out.numel() >= logits.size(0), "out must be int64 with >= S elements");
  check(kv_argmax(logits.data_ptr(), logits.stride(0), (int)logits.size(0), (int)logits.size(1), (long long*)out.data_ptr(), cur_stream()), "kv_argmax");
}

void prefetch(at::Tensor t, int64_t workgroups, at::Tensor sink) {
  CHECK_DEV(t); CHECK_CONTIG(t); CHECK_DEV(sink);
  TORCH_CHECK(sink.scalar_type() == at::kInt && sink.numel() >= 1, "sink must be an int32 tensor");
  TORCH_CHECK(((uintptr_t)t.data_ptr() & 15) == 0, "tensor must be 16-byte aligned");
  check(kv_prefetch(t.data_ptr(), (long long)t.numel() * t.element_size(), (int)workgroups, (unsigned int*)sink.data_ptr(), cur_stream()), "kv_prefetch");
}

// Temperature / top-k / top-p sampling of rows `rows` of bf16 logits [S, V]; out[rows[r]] = token id.
void sample_rows(at::Tensor logits, at::Tensor rows, at::Tensor temperature, at::Tensor top_k, at::Tensor top_p, at::Tensor uniform, at::Tensor out) {
  CHECK_DEV(logits); CHECK_BF16(logits); CHECK_DEV(rows); CHECK_DEV(temperature); CHECK_DEV(top_k); CHECK_DEV(top_p); CHECK_DEV(uniform); CHECK_DEV(out);
  CHECK_CONTIG(rows); CHECK_CONTIG(temperature); CHECK_CONTIG(top_k); CHECK_CONTIG(top_p); CHECK_CONTIG(uniform); CHECK_CONTIG(out);
  TORCH_CHECK(logits.dim() == 2 && logits.stride(1) == 1, "logits must be [S, V] with unit inner stride");
  const int64_t R = rows.numel();
  TORCH_CHECK(rows.scalar_type() == at::kLong && out.scalar_type() == at::kLong, "rows / out must be int64");
  TORCH_CHECK(temperature.scalar_type() == at::kFloat && top_p.scalar_type() == at::kFloat && uniform.scalar_type() == at::kFloat && top_k.scalar_type() == at::kInt,
              "temperature / top_p / uniform must be fp32, top_k int32");
  TORCH_CHECK(temperature.numel() == R && top_k.numel() == R && top_p.numel() == R && uniform.numel() == R, "one parameter per sampled row");
  // rows index both logits and out: the kernel trusts them, so their range is the caller's contract
  TORCH_CHECK(out.numel() >= logits.size(0), "out must have an entry for every logits row");
  check(kv_sample_rows(logits.data_ptr(), logits.stride(0), (int)logits.size(1), (const long long*)rows.data_ptr(), (int)R, (const float*)temperature.data_ptr(),
                       (const int*)top_k.data_ptr(), (const float*)top_p.data_ptr(), (const float*)uniform.data_ptr(), (long long*)out.data_ptr(), cur_stream()),
        "kv_sample_rows");
}

// the optional script tensors of guided decoding: script int32 [S, 2, L] (entries, rest), cursor int32 [S]
static const int* script_args(const c10::optional<at::Tensor>& script, const c10::optional<at::Tensor>& cursor, const at::Tensor& state, int** cur_ptr, int* L) {
  *cur_ptr = nullptr;
  *L = 0;
  if (!script.has_value()) return nullptr;
  TORCH_CHECK(cursor.has_value(), "a script needs its cursor tensor");
  const at::Tensor &sc = *script, &cu = *cursor;
  CHECK_DEV(sc); CHECK_CONTIG(sc); CHECK_DEV(cu); CHECK_CONTIG(cu);
  TORCH_CHECK(sc.scalar_type() == at::kInt && cu.scalar_type() == at::kInt, "script and cursor must be int32");
  TORCH_CHECK(sc.dim() == 3 && sc.size(0) == state.size(0) && sc.size(1) == 2 && sc.size(2) >= 1 && cu.numel() == state.size(0), "script is [S, 2, L] and cursor [S] for a state of [S, 4]");
  *cur_ptr = (int*)cu.data_ptr();
  *L = (int)sc.size(2);
  return (const int*)sc.data_ptr();
}

// Guided decoding (csrc/guided.hip).  rows: int64 logits rows, slots: int32 state rows, init: int32 [R, 4]
// (init[r][0] < 0 keeps the device state), state: int32 [S, 4], cls: uint8 class per token id; script /
// cursor / cur_init (optional): the skeleton scripts of schema-guided sequences (cur_init: int32 [R]).
void guided_mask(at::Tensor logits, at::Tensor rows, at::Tensor slots, at::Tensor init, at::Tensor state, at::Tensor cls, int64_t vc, c10::optional<at::Tensor> script,
                 c10::optional<at::Tensor> cursor, c10::optional<at::Tensor> cur_init) {
  CHECK_DEV(logits); CHECK_BF16(logits); CHECK_DEV(rows); CHECK_DEV(slots); CHECK_DEV(init); CHECK_DEV(state); CHECK_DEV(cls);
  CHECK_CONTIG(rows); CHECK_CONTIG(slots); CHECK_CONTIG(init); CHECK_CONTIG(state); CHECK_CONTIG(cls);
  TORCH_CHECK(logits.dim() == 2 && logits.stride(1) == 1, "logits must be [S, V] with unit inner stride");
  const int64_t R = rows.numel();
  TORCH_CHECK(rows.scalar_type() == at::kLong && slots.scalar_type() == at::kInt && init.scalar_type() == at::kInt && state.scalar_type() == at::kInt && cls.scalar_type() == at::kByte,
              "rows int64, slots / init / state int32, cls uint8");
  TORCH_CHECK(slots.numel() == R && init.numel() == 4 * R && state.dim() == 2 && state.size(1) == 4, "one slot and one init record per row; state is [S, 4]");
  TORCH_CHECK(cls.numel() <= logits.size(1), "more token classes than logits columns");
  int* curp = nullptr;
  int L = 0;
  const int* scp = script_args(script, cursor, state, &curp, &L);
  const int* cip = nullptr;
  if (scp != nullptr) {
    TORCH_CHECK(cur_init.has_value(), "a script needs cur_init");
    const at::Tensor& ci = *cur_init;
    CHECK_DEV(ci); CHECK_CONTIG(ci);
    TORCH_CHECK(ci.scalar_type() == at::kInt && ci.numel() == R, "cur_init must be int32 [R]");
    cip = (const int*)ci.data_ptr();
  }
  check(kv_guided_mask(logits.data_ptr(), logits.stride(0), (int)logits.size(1), (int)cls.numel(), (const long long*)rows.data_ptr(), (const int*)slots.data_ptr(),
                       (const int*)init.data_ptr(), (int)R, (int*)state.data_ptr(), (const unsigned char*)cls.data_ptr(), (int)vc, scp, curp, cip, L, cur_stream()),
        "kv_guided_mask");
}

void guided_advance(at::Tensor out_ids, at::Tensor rows, at::Tensor slots, at::Tensor state, at::Tensor cls, c10::optional<at::Tensor> script,
                    c10::optional<at::Tensor> cursor) {
  CHECK_DEV(out_ids); CHECK_DEV(rows); CHECK_DEV(slots); CHECK_DEV(state); CHECK_DEV(cls);
  CHECK_CONTIG(out_ids); CHECK_CONTIG(rows); CHECK_CONTIG(slots); CHECK_CONTIG(state); CHECK_CONTIG(cls);
  TORCH_CHECK(out_ids.scalar_type() == at::kLong && rows.scalar_type() == at::kLong && slots.scalar_type() == at::kInt && state.scalar_type() == at::kInt && cls.scalar_type() == at::kByte,
              "out_ids / rows int64, slots / state int32, cls uint8");
  TORCH_CHECK(slots.numel() == rows.numel() && state.dim() == 2 && state.size(1) == 4, "one slot per row; state is [S, 4]");
  int* curp = nullptr;
  int L = 0;
  const int* scp = script_args(script, cursor, state, &curp, &L);
  check(kv_guided_advance((const long long*)out_ids.data_ptr(), (const long long*)rows.data_ptr(), (const int*)slots.data_ptr(), (int)rows.numel(), (int*)state.data_ptr(),
                          (const unsigned char*)cls.data_ptr(), (int)cls.numel(), scp, curp, L, cur_stream()),
        "kv_guided_advance");
}

void scale_rows(at::Tensor x, at::Tensor row_ss, double eps, at::Tensor y) {
  CHECK_DEV(x); CHECK_BF16(x); CHECK_CONTIG(x); CHECK_DEV(y); CHECK_BF16(y); CHECK_CONTIG(y);
  TORCH_CHECK(x.dim() == 2 && y.sizes() == x.sizes() && x.size(1) % 8 == 0, "x, y must be bf16 [T, H] with H % 8 == 0");
  int parts = 0;
  const float* ssp = ss_ptr(row_ss, x.size(0), &parts);
  check(kv_scale_rows(x.data_ptr(), ssp, parts, (float)eps, y.data_ptr(), (int)x.size(0), (int)x.size(1), cur_stream()), "kv_scale_rows");
}

// slabs[0][t][:N] += bias[n] * sqrt(mean square of row t + eps) (row_ss: [T, parts] partial sums of squares over h_in
// columns, or None: the bias as it is) — see slab_add_bias_kernel.
void slab_add_bias(at::Tensor slabs, at::Tensor bias, c10::optional<at::Tensor> row_ss, int64_t h_in, double eps) {
  CHECK_DEV(slabs); CHECK_DEV(bias); CHECK_CONTIG(bias);
  TORCH_CHECK(slabs.scalar_type() == at::kFloat && slabs.dim() == 3 && slabs.stride(2) == 1, "slabs must be fp32 [S, T, N] with unit inner stride");
  TORCH_CHECK(bias.scalar_type() == at::kFloat && bias.numel() <= slabs.size(2), "bias must be fp32 with at most N entries");
  int parts = 0;
  const float* ssp = nullptr;
  if (row_ss.has_value()) {
    TORCH_CHECK(h_in > 0, "h_in must be positive with a pending row scale");
    ssp = ss_ptr(row_ss, slabs.size(1), &parts);
  }
  check(kv_slab_add_bias((float*)slabs.data_ptr(), slabs.stride(1), (const float*)bias.data_ptr(), ssp, parts, h_in > 0 ? 1.f / (float)h_in : 0.f, (float)eps, (int)slabs.size(1),
                         (int)bias.numel(), cur_stream()),
        "kv_slab_add_bias");
}

void swiglu_interleaved(at::Tensor gu, at::Tensor act) {
  CHECK_DEV(gu); CHECK_BF16(gu); CHECK_BF16(act); CHECK_CONTIG(gu); CHECK_CONTIG(act);
  TORCH_CHECK(gu.dim() == 2 && act.dim() == 2 && gu.size(0) == act.size(0) && gu.size(1) == 2 * act.size(1), "gu [T, 2I], act [T, I]");
  check(kv_swiglu_interleaved(gu.data_ptr(), act.data_ptr(), (int)gu.size(0), (int)act.size(1), cur_stream()), "kv_swiglu_interleaved");
}

at::Tensor pack_weight(at::Tensor W) {
  CHECK_DEV(W); CHECK_BF16(W); CHECK_CONTIG(W);
  TORCH_CHECK(W.dim() == 2, "W must be [N, K]");
  auto Wp = at::empty_like(W);
  check(kv_pack_weight(W.data_ptr(), Wp.data_ptr(), (int)W.size(0), (int)W.size(1), cur_stream()), "kv_pack_weight");
  return Wp;
}

// MXFP4 packed weights + E8M0 scale words -> `out`, bf16 in the fragment order of pack_weight
// (the first N * K elements of `out`, a scratch buffer that may be larger).
void unpack_mxfp4(at::Tensor Wp4, at::Tensor scales, at::Tensor out, int64_t N, int64_t K) {
  TORCH_CHECK(Wp4.is_cuda() && scales.is_cuda() && out.is_cuda(), "unpack_mxfp4: GPU tensors");
  TORCH_CHECK(Wp4.scalar_type() == at::kByte && scales.scalar_type() == at::kInt && out.scalar_type() == at::kBFloat16, "unpack_mxfp4: uint8 weights, int32 scales, bf16 output");
  TORCH_CHECK(Wp4.is_contiguous() && scales.is_contiguous() && out.is_contiguous(), "unpack_mxfp4: contiguous tensors");
  TORCH_CHECK(N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0, "unpack_mxfp4: N % 16 == 0, K % 128 == 0");
  TORCH_CHECK(Wp4.numel() == N * K / 2 && scales.numel() == N * K / 32 / 4 && out.numel() >= N * K, "unpack_mxfp4: sizes do not match N x K");
  check(kv_unpack_mxfp4(Wp4.data_ptr(), scales.data_ptr(), out.data_ptr(), (int)N, (int)K, cur_stream()), "kv_unpack_mxfp4");
}

at::Tensor pack_weight_fp8(at::Tensor W) {
  CHECK_DEV(W); CHECK_CONTIG(W);
  TORCH_CHECK(W.dim() == 2 && W.scalar_type() == at::kByte, "W must be uint8 [N, K] (e4m3 bytes)");
  TORCH_CHECK(W.size(0) % 16 == 0 && W.size(1) % 64 == 0, "N % 16 == 0 and K % 64 == 0 required");
  auto Wp = at::empty_like(W);
  check(kv_pack_weight_fp8(W.data_ptr(), Wp.data_ptr(), (int)W.size(0), (int)W.size(1), cur_stream()), "kv_pack_weight_fp8");
  return Wp;
}

// ---- weight-streaming GEMM v2 (packed weights, LDS-shared x, split-K slabs) -------------
std::pair<int64_t, int64_t> stream_gemm_plan(int64_t N, int64_t K, int64_t ks_per_chunk, int64_t max_splits) {
  int cps = 0;
  int splits = kv_stream_gemm_plan((int)N, (int)K, (int)ks_per_chunk, (int)max_splits, &cps);
  return {splits, cps};
}

// epi 0: out bf16 [M, N]; epi 1: out fp32 [splits, M, N]; epi 2: out bf16 [M, N/2] (SwiGLU).
const float* ss_ptr(const c10::optional<at::Tensor>& row_ss, int64_t rows, int* parts) {
  *parts = 0;
  if (!row_ss.has_value()) return nullptr;
  const at::Tensor& t = *row_ss;
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.dim() == 2 && t.size(0) >= rows, "row_ss must be contiguous fp32 [>=rows, parts]");
  *parts = (int)t.size(1);
  return (const float*)t.data_ptr();
}

// Wp: bf16 packed weight, or uint8 (OCP e4m3 bytes, kv_pack_weight_fp8 order) with w_scale fp32 [N].
void stream_gemm(at::Tensor Wp, int64_t N, int64_t K, at::Tensor x, at::Tensor out, int64_t epi, int64_t ks_per_chunk, int64_t splits, int64_t chunks_per_split,
                 c10::optional<at::Tensor> row_ss, double eps, c10::optional<at::Tensor> w_scale) {
  CHECK_DEV(Wp); CHECK_BF16(x); CHECK_CONTIG(Wp); CHECK_DEV(out);
  const float* wsp = nullptr;
  if (w_scale.has_value()) {
    const at::Tensor& ws = *w_scale;
    TORCH_CHECK(Wp.scalar_type() == at::kByte, "FP8 weights are passed as uint8 bytes");
    TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kFloat && ws.is_contiguous() && ws.numel() == N, "w_scale must be contiguous fp32 [N]");
    TORCH_CHECK(K % 64 == 0, "FP8 weights need K % 64 == 0");
    wsp = (const float*)ws.data_ptr();
  } else {
    CHECK_BF16(Wp);
  }
  TORCH_CHECK(Wp.numel() == N * K, "packed weight must hold N*K elements");
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && x.stride(1) == 1 && x.stride(0) % 8 == 0, "x must be [M, K] with 16-byte aligned rows");
  const int64_t M = x.size(0);
  int parts = 0;
  const float* ssp = ss_ptr(row_ss, M, &parts);
  long long ldy = 0, slab_stride = 0;
  if (epi == 1) {
    TORCH_CHECK(out.scalar_type() == at::kFloat && out.is_contiguous() && out.dim() == 3 && out.size(0) == splits && out.size(1) == M && out.size(2) == N,
                "slab output must be contiguous fp32 [splits, M, N]");
    ldy = N; slab_stride = M * N;
  } else {
    CHECK_BF16(out);
    TORCH_CHECK(out.dim() == 2 && out.size(0) == M && out.size(1) == (epi == 2 ? N / 2 : N) && out.stride(1) == 1 && out.stride(0) % 4 == 0, "bad output shape");
    TORCH_CHECK(splits == 1, "bf16 / SwiGLU epilogues need a single split");
    ldy = out.stride(0);
  }
  check(kv_stream_gemm(Wp.data_ptr(), x.data_ptr(), out.data_ptr(), (int)M, (int)N, (int)K, x.stride(0), ldy, (int)epi, (int)ks_per_chunk, (int)splits,
                       (int)chunks_per_split, slab_stride, ssp, parts, (float)eps, wsp, cur_stream()),
        "kv_stream_gemm");
}

// ---- weight-streaming GEMM v3 (deep register pipeline; bf16 or FP8 weights) ----------------------
std::pair<int64_t, int64_t> stream_gemm3_plan(int64_t N, int64_t K, int64_t nb, int64_t max_splits, int64_t target_wgs, int64_t nw) {
  int gps = 0;
  int splits = kv_stream_gemm3_plan((int)N, (int)K, (int)nb, (int)nw, (int)max_splits, (int)target_wgs, &gps);
  TORCH_CHECK(splits > 0, "stream_gemm3: K must be a multiple of 128*nb");
  return {splits, gps};
}

void stream_gemm3(at::Tensor Wp, int64_t N, int64_t K, at::Tensor x, at::Tensor out, int64_t epi, int64_t nb, int64_t splits, int64_t groups_per_split,
                  c10::optional<at::Tensor> row_ss, double eps, c10::optional<at::Tensor> w_scale, int64_t nw, int64_t weight_bits,
                  c10::optional<at::Tensor> residual, bool has_residual, c10::optional<at::Tensor> ss_out, c10::optional<at::Tensor> counters, int64_t ft) {
  CHECK_DEV(Wp); CHECK_BF16(x); CHECK_CONTIG(Wp); CHECK_DEV(out);
  TORCH_CHECK(ft == 1 || ft == 2, "ft (16-feature tiles per wave) must be 1 or 2");
  const float* wsp = nullptr;
  if (weight_bits == 4) {
    // MXFP4: two e2m1 values per byte, block scales as int32 [N/16, K/128, 16] (four E8M0 bytes each)
    TORCH_CHECK(w_scale.has_value(), "MXFP4 weights need their block scales");
    const at::Tensor& ws = *w_scale;
    TORCH_CHECK(Wp.scalar_type() == at::kByte && Wp.numel() * 2 == N * K, "MXFP4 weights are passed as N*K/2 uint8 bytes");
    TORCH_CHECK(K % 128 == 0 && ws.is_cuda() && ws.scalar_type() == at::kInt && ws.is_contiguous() && ws.numel() == (N / 16) * (K / 128) * 16, "block scales must be contiguous int32 [N/16, K/128, 16]");
    wsp = (const float*)ws.data_ptr();
  } else if (w_scale.has_value()) {
    const at::Tensor& ws = *w_scale;
    TORCH_CHECK(Wp.scalar_type() == at::kByte, "FP8 weights are passed as uint8 bytes");
    TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kFloat && ws.is_contiguous() && ws.numel() == N, "w_scale must be contiguous fp32 [N]");
    wsp = (const float*)ws.data_ptr();
    TORCH_CHECK(Wp.numel() == N * K, "packed weight must hold N*K elements");
  } else {
    CHECK_BF16(Wp);
    TORCH_CHECK(Wp.numel() == N * K, "packed weight must hold N*K elements");
  }
  TORCH_CHECK(weight_bits == 0 || weight_bits == 4 || weight_bits == 8 || weight_bits == 16, "weight_bits must be 0 (by w_scale), 4, 8 or 16");
  TORCH_CHECK(nw == 4 || nw == 6 || nw == 7 || nw == 8, "nw (waves per workgroup) must be 4, 6, 7 or 8");
  TORCH_CHECK(N % (16 * ft) == 0 && (N % (16 * nw * ft) == 0 || (ft == 1 && epi != 3)), "N must be a multiple of 16 * ft (of 16 * nw * ft with two tiles per wave or the residual epilogue)");
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && x.stride(1) == 1 && x.stride(0) % 8 == 0, "x must be [M, K] with 16-byte aligned rows");
  const int64_t M = x.size(0);
  int parts = 0;
  const float* ssp = ss_ptr(row_ss, M, &parts);
  long long ldy = 0, slab_stride = 0;
  void* resp = nullptr;
  float* ssop = nullptr;
  int* cntp = nullptr;
  if (epi == 1 || epi == 3) {
    TORCH_CHECK(out.scalar_type() == at::kFloat && out.is_contiguous() && out.dim() == 3 && out.size(0) == splits && out.size(1) == M && out.size(2) == N,
                "slab output must be contiguous fp32 [splits, M, N]");
    ldy = N; slab_stride = M * N;
    if (epi == 3) {
      // residual epilogue: residual bf16 [M, N] in/out, ss_out fp32 [M, N / (16 nw)], counters int32 (>= N / (16 nw), zero at rest)
      TORCH_CHECK(residual.has_value() && ss_out.has_value() && counters.has_value(), "the residual epilogue needs residual, ss_out and counters");
      const at::Tensor &rs = *residual, &so = *ss_out, &ct = *counters;
      CHECK_DEV(rs); CHECK_BF16(rs); CHECK_CONTIG(rs); CHECK_DEV(so); CHECK_CONTIG(so); CHECK_DEV(ct); CHECK_CONTIG(ct);
      const int64_t groups = N / (16 * nw);
      TORCH_CHECK(rs.dim() == 2 && rs.size(0) == M && rs.size(1) == N, "residual must be bf16 [M, N]");
      TORCH_CHECK(so.scalar_type() == at::kFloat && so.numel() == M * groups, "ss_out must be fp32 [M, N / (16 nw)]");
      TORCH_CHECK(ct.scalar_type() == at::kInt && ct.numel() >= groups, "counters must be int32 with one entry per feature group");
      TORCH_CHECK(M <= 64, "the residual epilogue serves up to 64 rows");
      resp = rs.data_ptr(); ssop = (float*)so.data_ptr(); cntp = (int*)ct.data_ptr();
    }
  } else {
    CHECK_BF16(out);
    TORCH_CHECK(out.dim() == 2 && out.size(0) == M && out.size(1) == (epi == 2 ? N / 2 : N) && out.stride(1) == 1 && out.stride(0) % 4 == 0, "bad output shape");
    TORCH_CHECK(splits == 1, "bf16 / SwiGLU epilogues need a single split");
    ldy = out.stride(0);
  }
  check(kv_stream_gemm3(Wp.data_ptr(), x.data_ptr(), out.data_ptr(), (int)M, (int)N, (int)K, x.stride(0), ldy, (int)epi, (int)nb, (int)nw, (int)splits,
                        (int)groups_per_split, slab_stride, ssp, parts, (float)eps, wsp, (int)weight_bits, resp, has_residual ? 1 : 0, ssop, cntp, (int)ft, cur_stream()),
        "kv_stream_gemm3");
}

// residual[M, N] += x[M, K] @ W^T in place (bf16), without slabs: one workgroup per 16-feature tile, K split over
// its waves.  Returns the per-row sums of squares of the new residual, one per tile: fp32 [M, N / 16].
at::Tensor stream_gemm_rowsum(at::Tensor Wp, int64_t N, int64_t K, at::Tensor x, at::Tensor residual, bool has_residual, c10::optional<at::Tensor> w_scale, int64_t nb) {
  CHECK_DEV(Wp); CHECK_CONTIG(Wp); CHECK_BF16(x); CHECK_BF16(residual); CHECK_CONTIG(residual);
  const float* wsp = nullptr;
  if (w_scale.has_value()) {
    const at::Tensor& ws = *w_scale;
    TORCH_CHECK(Wp.scalar_type() == at::kByte, "FP8 weights are passed as uint8 bytes");
    TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kFloat && ws.is_contiguous() && ws.numel() == N, "w_scale must be contiguous fp32 [N]");
    wsp = (const float*)ws.data_ptr();
  } else {
    CHECK_BF16(Wp);
  }
  TORCH_CHECK(Wp.numel() == N * K, "packed weight must hold N*K elements");
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && x.stride(1) == 1 && x.stride(0) % 8 == 0, "x must be [M, K] with 16-byte aligned rows");
  const int64_t M = x.size(0);
  TORCH_CHECK(M > 0 && M <= 64 && M % 16 == 0, "rows must be 16, 32, 48 or 64");
  TORCH_CHECK(N % 16 == 0 && K % 1024 == 0, "N must be a multiple of 16 and K of 1024");
  TORCH_CHECK(residual.dim() == 2 && residual.size(0) == M && residual.size(1) == N, "residual must be bf16 [M, N]");
  auto ss = at::empty({M, N / 16}, residual.options().dtype(at::kFloat));
  check(kv_stream_gemm_rowsum(Wp.data_ptr(), x.data_ptr(), (int)M, (int)N, (int)K, x.stride(0), (int)nb, residual.data_ptr(), has_residual ? 1 : 0, (float*)ss.data_ptr(),
                              wsp, cur_stream()),
        "kv_stream_gemm_rowsum");
  return ss;
}

// ---- persistent decode-layers kernel (csrc/engine_kernels.hip: decode_layers_kernel)
// Device memory the caches do not hold (hipDeviceMallocUncached), zeroed, as a uint8 tensor: what workgroups of ONE launch
// exchange across XCDs (each XCD has an L2 of its own) without cache write-backs or invalidates.
at::Tensor alloc_uncached(int64_t nbytes) {
  TORCH_CHECK(nbytes > 0, "alloc_uncached: size must be positive");
  void* p = nullptr;
  TORCH_CHECK(hipExtMallocWithFlags(&p, (size_t)nbytes, hipDeviceMallocUncached) == hipSuccess && p != nullptr, "hipExtMallocWithFlags(uncached) failed");
  TORCH_CHECK(hipMemset(p, 0, (size_t)nbytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess, "zeroing the uncached buffer failed");
  int dev = 0;
  (void)hipGetDevice(&dev);
  return at::from_blob(p, {nbytes}, [](void* q) { (void)hipFree(q); }, at::TensorOptions().dtype(at::kByte).device(at::kCUDA, dev));
}

int64_t decode_layers_workspace_bytes(int64_t H, int64_t nh, int64_t nkv, int64_t I, int64_t s_qkv, int64_t s_proj) {
  return kv_decode_layers_workspace_bytes((int)H, (int)nh, (int)nkv, (int)I, (int)s_qkv, (int)s_proj);
}
std::vector<int64_t> decode_layers_layout(int64_t H, int64_t nh, int64_t nkv, int64_t I, int64_t s_qkv, int64_t s_proj) {
  long long v[10];
  kv_decode_layers_layout((int)H, (int)nh, (int)nkv, (int)I, (int)s_qkv, (int)s_proj, v);
  return std::vector<int64_t>(v, v + 10);
}
int64_t decode_layers_state_offset(int64_t H, int64_t nh, int64_t nkv, int64_t I, int64_t s_qkv, int64_t s_proj) {
  return kv_decode_layers_state_offset((int)H, (int)nh, (int)nkv, (int)I, (int)s_qkv, (int)s_proj);
}

// layers: int64 [L, 6] device tensor of pointers (wqkv, wo, wgu, wdown, K cache, V cache per layer); plans: 20 ints (4 x N, K, waves,
// splits, chunks per split); residual bf16 [T, H] and ss fp32 [T, ceil(H / 512)] are updated in place; ws: alloc_uncached tensor.
void decode_layers(at::Tensor layers, std::vector<int64_t> plans, int64_t nh, int64_t nkv, int64_t I, at::Tensor positions, at::Tensor cos_t, at::Tensor sin_t,
                   at::Tensor slot_mapping, at::Tensor block_tables, at::Tensor tok_seq, at::Tensor ctx_lens, double eps, double scale, c10::optional<at::Tensor> kv_scale,
                   at::Tensor residual, at::Tensor ss, at::Tensor ws, int64_t grid, bool timeline) {
  CHECK_DEV(layers); CHECK_CONTIG(layers); CHECK_BF16(residual); CHECK_CONTIG(residual); CHECK_DEV(ss); CHECK_CONTIG(ss); CHECK_DEV(ws);
  CHECK_I32(positions); CHECK_I32(slot_mapping); CHECK_I32(block_tables); CHECK_I32(tok_seq); CHECK_I32(ctx_lens);
  CHECK_CONTIG(positions); CHECK_CONTIG(slot_mapping); CHECK_CONTIG(block_tables); CHECK_CONTIG(tok_seq); CHECK_CONTIG(ctx_lens); CHECK_CONTIG(cos_t); CHECK_CONTIG(sin_t);
  TORCH_CHECK(layers.scalar_type() == at::kLong && layers.dim() == 2 && layers.size(1) == 6, "layers must be int64 [L, 6]");
  TORCH_CHECK(plans.size() == 20, "plans must hold 4 x (N, K, waves, splits, chunks per split)");
  TORCH_CHECK(residual.dim() == 2 && ss.scalar_type() == at::kFloat, "residual must be [T, H], ss fp32");
  const int64_t T = residual.size(0), H = residual.size(1);
  TORCH_CHECK(ss.numel() == T * ((H + 511) / 512), "ss must be [T, ceil(H / 512)]");
  TORCH_CHECK(positions.numel() >= T && slot_mapping.numel() >= T && tok_seq.numel() >= T && ctx_lens.numel() >= T, "step inputs shorter than T");
  TORCH_CHECK(cos_t.scalar_type() == at::kFloat && sin_t.scalar_type() == at::kFloat && cos_t.dim() == 2 && cos_t.size(1) == 64 && sin_t.sizes() == cos_t.sizes(),
              "cos/sin must be float32 [max_pos, 64]");
  TORCH_CHECK(block_tables.dim() == 2, "block_tables must be [seqs, max_blocks]");
  int pl[20];
  for (int i = 0; i < 20; ++i) pl[i] = (int)plans[i];
  const int s_proj = pl[8] > pl[18] ? pl[8] : pl[18];
  TORCH_CHECK(ws.scalar_type() == at::kByte && ws.numel() >= kv_decode_layers_workspace_bytes((int)H, (int)nh, (int)nkv, (int)I, pl[3], s_proj), "workspace too small");
  const float* ksc = nullptr;
  if (kv_scale.has_value()) {
    const at::Tensor& k = *kv_scale;
    TORCH_CHECK(k.is_cuda() && k.scalar_type() == at::kFloat && k.is_contiguous() && k.numel() == layers.size(0) * 2 * nkv, "kv_scale must be fp32 [L, 2, nkv]");
    ksc = (const float*)k.data_ptr();
  }
  check(kv_decode_layers(layers.data_ptr(), (int)layers.size(0), (int)T, (int)H, (int)nh, (int)nkv, (int)I, pl, (const int*)positions.data_ptr(), (const float*)cos_t.data_ptr(),
                         (const float*)sin_t.data_ptr(), (const int*)slot_mapping.data_ptr(), (const int*)block_tables.data_ptr(), (int)block_tables.size(1),
                         (const int*)tok_seq.data_ptr(), (const int*)ctx_lens.data_ptr(), (float)eps, (float)scale, ksc, residual.data_ptr(), (float*)ss.data_ptr(),
                         ws.data_ptr(), (int)grid, timeline ? 1 : 0, cur_stream()),
        "kv_decode_layers");
}

// ---- tiled MFMA GEMM for wide token batches (csrc/wide_gemm.hip); same packed weights as the streaming kernels
int64_t wide_gemm_stage_ksteps(int64_t bm, int64_t bn, int64_t variant) { return kv_wide_gemm_stage_ksteps((int)bm, (int)bn, (int)variant); }

void wide_gemm(at::Tensor Wp, int64_t N, int64_t K, at::Tensor x, at::Tensor out, int64_t epi, int64_t bm, int64_t bn, int64_t variant, int64_t splits,
               int64_t stages_per_split, c10::optional<at::Tensor> row_ss, double eps, c10::optional<at::Tensor> w_scale, bool m_fastest) {
  CHECK_DEV(Wp); CHECK_BF16(x); CHECK_CONTIG(Wp); CHECK_DEV(out);
  const float* wsp = nullptr;
  if (w_scale.has_value()) {
    const at::Tensor& ws = *w_scale;
    TORCH_CHECK(Wp.scalar_type() == at::kByte, "FP8 weights are passed as uint8 bytes");
    TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kFloat && ws.is_contiguous() && ws.numel() == N, "w_scale must be contiguous fp32 [N]");
    wsp = (const float*)ws.data_ptr();
  } else {
    CHECK_BF16(Wp);
  }
  TORCH_CHECK(Wp.numel() == N * K, "packed weight must hold N*K elements");
  TORCH_CHECK(K % 64 == 0 && N % bn == 0, "wide_gemm: K must be a multiple of 64 and N of the feature tile");
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && x.stride(1) == 1 && x.stride(0) % 8 == 0, "x must be [M, K] with 16-byte aligned rows");
  const int64_t M = x.size(0);
  int parts = 0;
  const float* ssp = ss_ptr(row_ss, M, &parts);
  long long ldy = 0, slab_stride = 0;
  if (epi == 1) {
    TORCH_CHECK(out.scalar_type() == at::kFloat && out.is_contiguous() && out.dim() == 3 && out.size(0) == splits && out.size(1) == M && out.size(2) == N,
                "slab output must be contiguous fp32 [splits, M, N]");
    ldy = N; slab_stride = M * N;
  } else {
    CHECK_BF16(out);
    TORCH_CHECK(out.dim() == 2 && out.size(0) == M && out.size(1) == (epi == 2 ? N / 2 : N) && out.stride(1) == 1 && out.stride(0) % 4 == 0, "bad output shape");
    TORCH_CHECK(splits == 1, "bf16 / SwiGLU epilogues need a single split");
    ldy = out.stride(0);
  }
  check(kv_wide_gemm(Wp.data_ptr(), x.data_ptr(), out.data_ptr(), (int)M, (int)N, (int)K, x.stride(0), ldy, (int)epi, (int)bm, (int)bn, (int)variant, (int)splits,
                     (int)stages_per_split, slab_stride, ssp, parts, (float)eps, wsp, m_fastest ? 1 : 0, cur_stream()),
        "kv_wide_gemm");
}

// FP8 weights x FP8 activations on the block-scaled MFMA.  x8: uint8 [M, K] from quantize_rows_fp8, a_scale fp32 [M], w_scale fp32 [N].
void wide_gemm_fp8a(at::Tensor Wp, int64_t N, int64_t K, at::Tensor x8, at::Tensor a_scale, at::Tensor out, int64_t epi, int64_t bm, int64_t bn, int64_t variant,
                    int64_t splits, int64_t stages_per_split, c10::optional<at::Tensor> row_ss, double eps, at::Tensor w_scale, bool m_fastest) {
  CHECK_DEV(Wp); CHECK_CONTIG(Wp); CHECK_DEV(out); CHECK_DEV(x8); CHECK_DEV(a_scale); CHECK_DEV(w_scale);
  TORCH_CHECK(Wp.scalar_type() == at::kByte && x8.scalar_type() == at::kByte, "FP8 weights and activations are passed as uint8 bytes");
  TORCH_CHECK(w_scale.scalar_type() == at::kFloat && w_scale.is_contiguous() && w_scale.numel() == N, "w_scale must be contiguous fp32 [N]");
  TORCH_CHECK(Wp.numel() == N * K, "packed weight must hold N*K elements");
  TORCH_CHECK(K % 128 == 0 && N % bn == 0, "wide_gemm_fp8a: K must be a multiple of 128 and N of the feature tile");
  TORCH_CHECK(x8.dim() == 2 && x8.size(1) == K && x8.stride(1) == 1 && x8.stride(0) % 16 == 0, "x8 must be [M, K] bytes with 16-byte aligned rows");
  const int64_t M = x8.size(0);
  TORCH_CHECK(a_scale.scalar_type() == at::kFloat && a_scale.is_contiguous() && a_scale.numel() == M, "a_scale must be contiguous fp32 [M]");
  int parts = 0;
  const float* ssp = ss_ptr(row_ss, M, &parts);
  long long ldy = 0, slab_stride = 0;
  if (epi == 1) {
    TORCH_CHECK(out.scalar_type() == at::kFloat && out.is_contiguous() && out.dim() == 3 && out.size(0) == splits && out.size(1) == M && out.size(2) == N,
                "slab output must be contiguous fp32 [splits, M, N]");
    ldy = N; slab_stride = M * N;
  } else {
    CHECK_BF16(out);
    TORCH_CHECK(out.dim() == 2 && out.size(0) == M && out.size(1) == (epi == 2 ? N / 2 : N) && out.stride(1) == 1 && out.stride(0) % 4 == 0, "bad output shape");
    TORCH_CHECK(splits == 1, "bf16 / SwiGLU epilogues need a single split");
    ldy = out.stride(0);
  }
  check(kv_wide_gemm_fp8a(Wp.data_ptr(), x8.data_ptr(), out.data_ptr(), (int)M, (int)N, (int)K, x8.stride(0), ldy, (int)epi, (int)bm, (int)bn, (int)variant, (int)splits,
                          (int)stages_per_split, slab_stride, ssp, parts, (float)eps, (const float*)w_scale.data_ptr(), (const float*)a_scale.data_ptr(), m_fastest ? 1 : 0,
                          cur_stream()),
        "kv_wide_gemm_fp8a");
}

// MXFP4 weights (pack_mxfp4 layout + block scale words) x FP8 activations (natural k order) on the block-scaled MFMA
void wide_gemm_w4a8(at::Tensor Wp4, at::Tensor w_blk, int64_t N, int64_t K, at::Tensor x8, at::Tensor a_scale, at::Tensor out, int64_t epi, int64_t bm, int64_t bn, int64_t variant,
                    int64_t splits, int64_t stages_per_split, c10::optional<at::Tensor> row_ss, double eps, bool m_fastest) {
  CHECK_DEV(Wp4); CHECK_CONTIG(Wp4); CHECK_DEV(w_blk); CHECK_CONTIG(w_blk); CHECK_DEV(out); CHECK_DEV(x8); CHECK_DEV(a_scale);
  TORCH_CHECK(Wp4.scalar_type() == at::kByte && x8.scalar_type() == at::kByte && w_blk.scalar_type() == at::kInt, "MXFP4 weights / FP8 activations as uint8, block scales as int32");
  TORCH_CHECK(K % 128 == 0 && N % bn == 0 && Wp4.numel() == N * K / 2 && w_blk.numel() == N * (K / 128), "wide_gemm_w4a8: packed weights [N/16, K/128, 64, 16] and scales [N/16, K/128, 16]");
  TORCH_CHECK(x8.dim() == 2 && x8.size(1) == K && x8.stride(1) == 1 && x8.stride(0) % 16 == 0, "x8 must be [M, K] bytes with 16-byte aligned rows");
  const int64_t M = x8.size(0);
  TORCH_CHECK(a_scale.scalar_type() == at::kFloat && a_scale.is_contiguous() && a_scale.numel() == M, "a_scale must be contiguous fp32 [M]");
  int parts = 0;
  const float* ssp = ss_ptr(row_ss, M, &parts);
  long long ldy = 0, slab_stride = 0;
  if (epi == 1) {
    TORCH_CHECK(out.scalar_type() == at::kFloat && out.is_contiguous() && out.dim() == 3 && out.size(0) == splits && out.size(1) == M && out.size(2) == N,
                "slab output must be contiguous fp32 [splits, M, N]");
    ldy = N; slab_stride = M * N;
  } else {
    CHECK_BF16(out);
    TORCH_CHECK(out.dim() == 2 && out.size(0) == M && out.size(1) == (epi == 2 ? N / 2 : N) && out.stride(1) == 1 && out.stride(0) % 4 == 0, "bad output shape");
    TORCH_CHECK(splits == 1, "bf16 / SwiGLU epilogues need a single split");
    ldy = out.stride(0);
  }
  check(kv_wide_gemm_w4a8(Wp4.data_ptr(), (const int*)w_blk.data_ptr(), x8.data_ptr(), out.data_ptr(), (int)M, (int)N, (int)K, x8.stride(0), ldy, (int)epi, (int)bm, (int)bn, (int)variant,
                          (int)splits, (int)stages_per_split, slab_stride, ssp, parts, (float)eps, (const float*)a_scale.data_ptr(), m_fastest ? 1 : 0, cur_stream()),
        "kv_wide_gemm_w4a8");
}

// x bf16 [M, K] (rows may be strided) -> (x8 uint8 [M, K] in the packed weights' k order, per-token scale fp32 [M])
std::vector<at::Tensor> quantize_rows_fp8(at::Tensor x, bool permute) {
  CHECK_BF16(x);
  TORCH_CHECK(x.dim() == 2 && x.stride(1) == 1 && x.stride(0) % 8 == 0 && x.size(1) % 64 == 0, "x must be bf16 [M, K], K a multiple of 64, 16-byte aligned rows");
  const int64_t M = x.size(0), K = x.size(1);
  at::Tensor x8 = at::empty({M, K}, x.options().dtype(at::kByte));
  at::Tensor sc = at::empty({M}, x.options().dtype(at::kFloat));
  check(kv_quantize_rows_fp8(x.data_ptr(), x.stride(0), (int)M, (int)K, x8.data_ptr(), (float*)sc.data_ptr(), permute ? 1 : 0, cur_stream()), "kv_quantize_rows_fp8");
  return {x8, sc};
}

// ONE v_mfma_scale_f32_16x16x128_f8f6f4 on caller-supplied operand registers (int32 [64, 8] each, scale dwords int32 [64]) -> fp32 [64, 4]
at::Tensor mfma_scale_probe(at::Tensor a, at::Tensor b, at::Tensor sa, at::Tensor sb, int64_t fmt_a, int64_t fmt_b) {
  CHECK_DEV(a); CHECK_DEV(b); CHECK_DEV(sa); CHECK_DEV(sb); CHECK_CONTIG(a); CHECK_CONTIG(b); CHECK_CONTIG(sa); CHECK_CONTIG(sb);
  TORCH_CHECK(a.scalar_type() == at::kInt && b.scalar_type() == at::kInt && sa.scalar_type() == at::kInt && sb.scalar_type() == at::kInt, "int32 operands");
  TORCH_CHECK(a.numel() == 512 && b.numel() == 512 && sa.numel() == 64 && sb.numel() == 64, "operands are [64, 8], scales [64]");
  at::Tensor out = at::empty({64, 4}, a.options().dtype(at::kFloat));
  check(kv_mfma_scale_probe((const int*)a.data_ptr(), (const int*)b.data_ptr(), (const int*)sa.data_ptr(), (const int*)sb.data_ptr(), (float*)out.data_ptr(), (int)fmt_a, (int)fmt_b,
                            cur_stream()),
        "kv_mfma_scale_probe");
  return out;
}

void add_rmsnorm_slabs(at::Tensor slabs, at::Tensor residual, at::Tensor w, at::Tensor y, double eps) {
  CHECK_DEV(slabs); CHECK_CONTIG(slabs); CHECK_BF16(residual); CHECK_BF16(w); CHECK_BF16(y); CHECK_CONTIG(residual); CHECK_CONTIG(w); CHECK_CONTIG(y);
  TORCH_CHECK(slabs.scalar_type() == at::kFloat && slabs.dim() == 3, "slabs must be fp32 [S, T, H]");
  const int64_t T = slabs.size(1), H = slabs.size(2);
  TORCH_CHECK(residual.dim() == 2 && residual.size(0) == T && residual.size(1) == H && y.sizes() == residual.sizes() && w.numel() == H, "shape mismatch");
  TORCH_CHECK(H % 8 == 0 && H <= 32768, "hidden size must be a multiple of 8 and <= 32768");
  check(kv_add_rmsnorm_slabs((const float*)slabs.data_ptr(), (int)slabs.size(0), T * H, residual.data_ptr(), w.data_ptr(), y.data_ptr(), (int)T, (int)H, (float)eps,
                             cur_stream()),
        "kv_add_rmsnorm_slabs");
}

void rope_kv_write_slabs(at::Tensor slabs, at::Tensor q_out, at::Tensor positions, at::Tensor cos_t, at::Tensor sin_t, at::Tensor k_cache, at::Tensor v_cache,
                         at::Tensor slot_mapping, int64_t nh, int64_t nkv, int64_t hd, c10::optional<at::Tensor> row_ss, int64_t h_in, double eps,
                         c10::optional<at::Tensor> kv_scale) {
  CHECK_DEV(slabs); CHECK_CONTIG(slabs); CHECK_BF16(q_out); CHECK_CONTIG(q_out); CHECK_I32(positions); CHECK_I32(slot_mapping);
  const float* ksc = check_cache(k_cache, v_cache, kv_scale);
  CHECK_CONTIG(cos_t); CHECK_CONTIG(sin_t); CHECK_CONTIG(positions); CHECK_CONTIG(slot_mapping);
  TORCH_CHECK(slabs.scalar_type() == at::kFloat && slabs.dim() == 3 && slabs.size(2) == (nh + 2 * nkv) * hd, "slabs must be fp32 [S, T, (nh + 2 nkv) * hd]");
  const int64_t T = slabs.size(1);
  TORCH_CHECK(q_out.numel() == T * nh * hd, "q_out must hold T*nh*hd elements");
  int parts = 0;
  const float* ssp = ss_ptr(row_ss, T, &parts);
  TORCH_CHECK(ssp == nullptr || h_in > 0, "h_in (input width of the projection) is required with row_ss");
  TORCH_CHECK(positions.numel() >= T && slot_mapping.numel() >= T, "positions/slot_mapping shorter than T");
  TORCH_CHECK(cos_t.scalar_type() == at::kFloat && sin_t.scalar_type() == at::kFloat && cos_t.dim() == 2 && cos_t.size(1) == hd / 2 && sin_t.sizes() == cos_t.sizes(),
              "cos/sin must be float32 [max_pos, hd/2]");
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(1) == nkv && k_cache.size(3) == hd && v_cache.sizes() == k_cache.sizes(), "cache must be [blocks, nkv, block_size, hd]");
  check(kv_rope_kv_write_slabs((const float*)slabs.data_ptr(), (int)slabs.size(0), T * slabs.size(2), q_out.data_ptr(), (const int*)positions.data_ptr(),
                               (const float*)cos_t.data_ptr(), (const float*)sin_t.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                               (const int*)slot_mapping.data_ptr(), (int)T, (int)nh, (int)nkv, (int)hd, (int)k_cache.size(2), ssp, parts, (int)h_in, (float)eps,
                               ksc, cur_stream()),
        "kv_rope_kv_write_slabs");
}

// residual' = residual + x (x: bf16 [T, H] or fp32 slabs [S, T, H]); returns the per-row
// partial sums of squares [T, ceil(H/512)] of residual'.
at::Tensor residual_reduce(at::Tensor x, at::Tensor residual, bool has_residual, c10::optional<at::Tensor> pf_w, c10::optional<std::vector<int64_t>> pf_plan,
                           c10::optional<at::Tensor> pf_sink) {
  CHECK_DEV(x); CHECK_CONTIG(x); CHECK_BF16(residual); CHECK_CONTIG(residual);
  long long pfv[8];
  unsigned int* pfs = nullptr;
  const void* pfb = prefetch_args(pf_w, pf_plan, pf_sink, pfv, &pfs);
  TORCH_CHECK(residual.dim() == 2, "residual must be [T, H]");
  const int64_t T = residual.size(0), H = residual.size(1);
  TORCH_CHECK(H % 8 == 0, "hidden size must be a multiple of 8");
  auto ss = at::empty({T, (H + 511) / 512}, residual.options().dtype(at::kFloat));
  if (x.dim() == 3) {
    TORCH_CHECK(x.scalar_type() == at::kFloat && x.size(1) == T && x.size(2) == H && x.size(0) >= 1, "slabs must be fp32 [S, T, H]");
    check(kv_residual_reduce((const float*)x.data_ptr(), (int)x.size(0), T * H, nullptr, residual.data_ptr(), has_residual ? 1 : 0, (float*)ss.data_ptr(), (int)T, (int)H,
                             pfb, pfb ? pfv : nullptr, pfs, cur_stream()),
          "kv_residual_reduce");
  } else {
    CHECK_BF16(x);
    TORCH_CHECK(x.dim() == 2 && x.size(0) == T && x.size(1) == H, "x must be bf16 [T, H]");
    check(kv_residual_reduce(nullptr, 0, 0, x.data_ptr(), residual.data_ptr(), has_residual ? 1 : 0, (float*)ss.data_ptr(), (int)T, (int)H, pfb, pfb ? pfv : nullptr, pfs,
                             cur_stream()),
          "kv_residual_reduce");
  }
  return ss;
}

// ---- xGMI peer-memory collectives (csrc/xgmi_collectives.hip); the communicator is an opaque integer handle
int64_t xgmi_create(int64_t world, int64_t rank, int64_t slot_bytes) {
  void* h = kvx_create((int)world, (int)rank, slot_bytes);
  TORCH_CHECK(h != nullptr, "xgmi_create failed (world 1..8, slot bytes a multiple of 4096, uncached device memory available)");
  return (int64_t)(uintptr_t)h;
}
py::bytes xgmi_handle(int64_t h) {
  std::string buf((size_t)kvx_handle_bytes(), '\0');
  TORCH_CHECK(kvx_get_handle((void*)(uintptr_t)h, buf.data()) == 0, "hipIpcGetMemHandle failed");
  return py::bytes(buf);
}
void xgmi_open_peer(int64_t h, int64_t r, const std::string& handle) {
  TORCH_CHECK((int)handle.size() == kvx_handle_bytes(), "bad IPC handle size");
  const int rc = kvx_open_peer((void*)(uintptr_t)h, (int)r, handle.data());
  TORCH_CHECK(rc == 0, "hipIpcOpenMemHandle failed for peer ", r, " (rc ", rc, ")");
}
bool xgmi_ready(int64_t h) { return kvx_ready((void*)(uintptr_t)h) != 0; }
int64_t xgmi_errors(int64_t h) { return kvx_errors((void*)(uintptr_t)h); }
int64_t xgmi_calls(int64_t h) { return kvx_calls((void*)(uintptr_t)h); }
int64_t xgmi_slot_bytes(int64_t h) { return kvx_slot_bytes((void*)(uintptr_t)h); }
int64_t xgmi_max_wg() { return kvx_max_wg(); }
void xgmi_destroy(int64_t h) { kvx_destroy((void*)(uintptr_t)h); }

void xgmi_allgather(int64_t h, at::Tensor x, at::Tensor out) {
  CHECK_DEV(x); CHECK_DEV(out); CHECK_CONTIG(x); CHECK_CONTIG(out);
  const int64_t nbytes = x.numel() * x.element_size();
  TORCH_CHECK(nbytes % 16 == 0 && out.numel() * out.element_size() % nbytes == 0, "all-gather payload must be a multiple of 16 bytes, out = world x payload");
  check(kvx_allgather((void*)(uintptr_t)h, x.data_ptr(), out.data_ptr(), nbytes, cur_stream()), "kvx_allgather");
}
void xgmi_allreduce(int64_t h, at::Tensor x, at::Tensor out, bool two_shot) {
  CHECK_BF16(x); CHECK_BF16(out); CHECK_CONTIG(x); CHECK_CONTIG(out);
  TORCH_CHECK(x.numel() == out.numel(), "shape mismatch");
  check(kvx_allreduce((void*)(uintptr_t)h, x.data_ptr(), out.data_ptr(), x.numel(), two_shot ? 1 : 0, cur_stream()), "kvx_allreduce");
}
// residual' = residual + all-reduce(x) (x: bf16 [T, H] or fp32 slabs [S, T, H]); returns the partial sums of squares [T, ceil(H/512)]
at::Tensor xgmi_residual_allreduce(int64_t h, at::Tensor x, at::Tensor residual, bool has_residual) {
  CHECK_DEV(x); CHECK_CONTIG(x); CHECK_BF16(residual); CHECK_CONTIG(residual);
  TORCH_CHECK(residual.dim() == 2, "residual must be [T, H]");
  const int64_t T = residual.size(0), H = residual.size(1);
  auto ss = at::empty({T, (H + 511) / 512}, residual.options().dtype(at::kFloat));
  if (x.dim() == 3) {
    TORCH_CHECK(x.scalar_type() == at::kFloat && x.size(1) == T && x.size(2) == H, "slabs must be fp32 [S, T, H]");
    check(kvx_residual_allreduce((void*)(uintptr_t)h, (const float*)x.data_ptr(), (int)x.size(0), T * H, nullptr, residual.data_ptr(), has_residual ? 1 : 0,
                                 (float*)ss.data_ptr(), (int)T, (int)H, cur_stream()),
          "kvx_residual_allreduce");
  } else {
    CHECK_BF16(x);
    TORCH_CHECK(x.dim() == 2 && x.size(0) == T && x.size(1) == H, "x must be bf16 [T, H]");
    check(kvx_residual_allreduce((void*)(uintptr_t)h, nullptr, 0, 0, x.data_ptr(), residual.data_ptr(), has_residual ? 1 : 0, (float*)ss.data_ptr(), (int)T, (int)H,
                                 cur_stream()),
          "kvx_residual_allreduce");
  }
  return ss;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "kvmini gfx950 kernels";
  m.def("stream_state_bytes", &stream_state_bytes);
  m.def("radix_state_bytes", &radix_state_bytes);
  m.def("stream_hist", &stream_hist);
  m.def("token_gaps", &token_gaps);
  m.def("radix_select", &radix_select);
  m.def("radix_num_passes", &radix_num_passes);
  m.def("radix_init", &radix_init);
  m.def("radix_pass_hist", &radix_pass_hist);
  m.def("radix_pass_scan", &radix_pass_scan);
  m.def("equal_hist", &equal_hist);
  m.def("energy_trapz", &energy_trapz);
  m.def("add_rmsnorm", &add_rmsnorm);
  m.def("rope_kv_write", &rope_kv_write, py::arg("qkv"), py::arg("positions"), py::arg("cos_t"), py::arg("sin_t"), py::arg("k_cache"), py::arg("v_cache"),
        py::arg("slot_mapping"), py::arg("nh"), py::arg("nkv"), py::arg("hd"), py::arg("kv_scale") = py::none());
  m.def("paged_attention", &paged_attention, py::arg("q"), py::arg("k_cache"), py::arg("v_cache"), py::arg("block_tables"), py::arg("tok_seq"), py::arg("ctx_lens"),
        py::arg("out"), py::arg("scale"), py::arg("kv_scale") = py::none());
  m.def("prefill_attention", &prefill_attention, py::arg("q"), py::arg("k_cache"), py::arg("v_cache"), py::arg("block_tables"), py::arg("tiles"), py::arg("out"),
        py::arg("scale"), py::arg("kv_scale") = py::none(), py::arg("quads") = false);
  m.def("prefill_attention_split", &prefill_attention_split);
  m.def("split_decode_attention", &split_decode_attention);
  m.def("decode_attention_fused", &decode_attention_fused, py::arg("slabs"), py::arg("positions"), py::arg("cos_t"), py::arg("sin_t"), py::arg("k_cache"),
        py::arg("v_cache"), py::arg("slot_mapping"), py::arg("block_tables"), py::arg("tok_seq"), py::arg("ctx_lens"), py::arg("out"), py::arg("nh"),
        py::arg("row_ss"), py::arg("h_in"), py::arg("eps"), py::arg("scale"), py::arg("waves") = 4, py::arg("kv_scale") = py::none(), py::arg("n_split") = 1,
        py::arg("split_ws") = py::none(), py::arg("pf_w") = py::none(), py::arg("pf_plan") = py::none(), py::arg("pf_sink") = py::none());
  m.def("argmax", &argmax);
  m.def("embed_fed", &embed_fed);
  m.def("slab_add_bias", &slab_add_bias, py::arg("slabs"), py::arg("bias"), py::arg("row_ss") = py::none(), py::arg("h_in") = 0, py::arg("eps") = 0.0);
  m.def("slab_add_bias", &slab_add_bias, py::arg("slabs"), py::arg("bias"), py::arg("row_ss") = py::none(), py::arg("h_in") = 0, py::arg("eps") = 0.0);
  m.def("swiglu_interleaved", &swiglu_interleaved);
  m.def("scale_rows", &scale_rows);
  m.def("sample_rows", &sample_rows);
  m.def("guided_mask", &guided_mask, py::arg("logits"), py::arg("rows"), py::arg("slots"), py::arg("init"), py::arg("state"), py::arg("cls"), py::arg("vc"),
        py::arg("script") = py::none(), py::arg("cursor") = py::none(), py::arg("cur_init") = py::none());
  m.def("guided_advance", &guided_advance, py::arg("out_ids"), py::arg("rows"), py::arg("slots"), py::arg("state"), py::arg("cls"), py::arg("script") = py::none(),
        py::arg("cursor") = py::none());
  m.def("pack_weight", &pack_weight);
  m.def("stream_gemm_plan", &stream_gemm_plan);
  m.def("stream_gemm", &stream_gemm, py::arg("Wp"), py::arg("N"), py::arg("K"), py::arg("x"), py::arg("out"), py::arg("epi"), py::arg("ks_per_chunk"), py::arg("splits"),
        py::arg("chunks_per_split"), py::arg("row_ss") = py::none(), py::arg("eps") = 0.0, py::arg("w_scale") = py::none());
  m.def("pack_weight_fp8", &pack_weight_fp8);
  m.def("unpack_mxfp4", &unpack_mxfp4);
  m.def("stream_gemm3_plan", &stream_gemm3_plan, py::arg("N"), py::arg("K"), py::arg("nb"), py::arg("max_splits"), py::arg("target_wgs") = 512, py::arg("nw") = 4);
  m.def("stream_gemm3", &stream_gemm3, py::arg("Wp"), py::arg("N"), py::arg("K"), py::arg("x"), py::arg("out"), py::arg("epi"), py::arg("nb"), py::arg("splits"),
        py::arg("groups_per_split"), py::arg("row_ss") = py::none(), py::arg("eps") = 0.0, py::arg("w_scale") = py::none(), py::arg("nw") = 4, py::arg("weight_bits") = 0,
        py::arg("residual") = py::none(), py::arg("has_residual") = true, py::arg("ss_out") = py::none(), py::arg("counters") = py::none(), py::arg("ft") = 1);
  m.def("stream_gemm_rowsum", &stream_gemm_rowsum, py::arg("Wp"), py::arg("N"), py::arg("K"), py::arg("x"), py::arg("residual"), py::arg("has_residual"),
        py::arg("w_scale") = py::none(), py::arg("nb") = 2);
  m.def("alloc_uncached", &alloc_uncached);
  m.def("decode_layers_workspace_bytes", &decode_layers_workspace_bytes);
  m.def("decode_layers_state_offset", &decode_layers_state_offset);
  m.def("decode_layers_layout", &decode_layers_layout);
  m.def("decode_layers", &decode_layers);
  m.def("wide_gemm_stage_ksteps", &wide_gemm_stage_ksteps);
  m.def("wide_gemm_fp8a", &wide_gemm_fp8a);
  m.def("quantize_rows_fp8", &quantize_rows_fp8, py::arg("x"), py::arg("permute") = true);
  m.def("wide_gemm_w4a8", &wide_gemm_w4a8);
  m.def("mfma_scale_probe", &mfma_scale_probe);
  m.def("set_attention_mfma", [](bool on) { kv_set_attention_mfma(on ? 1 : 0); });
  m.def("set_attention_mfma_unsplit", [](bool on) { kv_set_attention_mfma_unsplit(on ? 1 : 0); });
  m.def("get_attention_mfma", []() { return kv_get_attention_mfma() != 0; });
  m.def("wide_gemm", &wide_gemm, py::arg("Wp"), py::arg("N"), py::arg("K"), py::arg("x"), py::arg("out"), py::arg("epi"), py::arg("bm"), py::arg("bn"), py::arg("variant") = 0,
        py::arg("splits") = 1, py::arg("stages_per_split") = 0, py::arg("row_ss") = py::none(), py::arg("eps") = 0.0, py::arg("w_scale") = py::none(), py::arg("m_fastest") = true);
  m.def("xgmi_create", &xgmi_create);
  m.def("xgmi_handle", &xgmi_handle);
  m.def("xgmi_open_peer", &xgmi_open_peer);
  m.def("xgmi_ready", &xgmi_ready);
  m.def("xgmi_errors", &xgmi_errors);
  m.def("xgmi_calls", &xgmi_calls);
  m.def("xgmi_slot_bytes", &xgmi_slot_bytes);
  m.def("xgmi_max_wg", &xgmi_max_wg);
  m.def("xgmi_destroy", &xgmi_destroy);
  m.def("xgmi_allgather", &xgmi_allgather);
  m.def("xgmi_allreduce", &xgmi_allreduce);
  m.def("xgmi_residual_allreduce", &xgmi_residual_allreduce);
  m.def("prefetch", &prefetch);
  m.def("add_rmsnorm_slabs", &add_rmsnorm_slabs);
  m.def("rope_kv_write_slabs", &rope_kv_write_slabs, py::arg("slabs"), py::arg("q_out"), py::arg("positions"), py::arg("cos_t"), py::arg("sin_t"), py::arg("k_cache"),
        py::arg("v_cache"), py::arg("slot_mapping"), py::arg("nh"), py::arg("nkv"), py::arg("hd"), py::arg("row_ss"), py::arg("h_in"), py::arg("eps"),
        py::arg("kv_scale") = py::none());
  m.def("residual_reduce", &residual_reduce, py::arg("x"), py::arg("residual"), py::arg("has_residual"), py::arg("pf_w") = py::none(), py::arg("pf_plan") = py::none(),
        py::arg("pf_sink") = py::none());
}
