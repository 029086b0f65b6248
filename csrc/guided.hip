// Guided JSON decoding on the device (gfx950): the automaton of kvmini/engine/guided.py.
//
// guided_mask_kernel     one 1024-thread workgroup per guided logits row: (re)initialise the
//                        sequence's automaton if the host says so, work out the set of token
//                        CLASSES that may come next, and overwrite the logit of every token of
//                        another class with -inf (bf16), 8 logits + 8 class bytes per trip.
// guided_advance_kernel  one thread per guided row: feed the token that argmax / sampling picked
//                        from the masked row to the automaton.
// Both run behind the forward pass on the engine's stream, so a step that is launched before the
// previous step's token has reached the host still sees the right automaton state.
//
// State of a sequence: int32[4] = (state, depth, container stack bits (1 = array), tokens left).
//
// Scripts (JSON schemas): a sequence may carry a script, int32 [2][L] = (entries, rest), and a
// cursor into it.  Entry >= 0: a literal token of the document's skeleton (a required key with its
// punctuation, the closing brace), FORCED — the mask leaves only that token — once the value in
// front of it is complete at the top level; -2 - mask: a value slot whose first token is
// restricted to the class set; -1: end; JMP_BASE - t: continue at entry t; CHOICE_BASE - n: the
// MODEL chooses among n alternatives — 2 n entries (key, target) follow, key >= 0 a first token,
// key a slot code a first value — which is what enum values, anyOf alternatives and optional
// array items compile to: the mask leaves the alternatives' tokens (and classes), the pick moves
// the cursor.  rest[i] = fewest tokens from entry i to the end of the sequence, which keeps the
// "can still be closed within the tokens left" rule (an alternative is offered only if it fits).
// The automaton is SET at the end of a literal (merged tokens such as '":' have no class to
// follow).  kvmini/engine/guided.py::allowed_scripted / step_scripted are the reference.

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

enum { C_OTHER, C_LBRACE, C_RBRACE, C_LBRACK, C_RBRACK, C_COLON, C_COMMA, C_QUOTE, C_STR, C_DIGIT_NZ, C_DIGIT, C_LIT, C_EOS, C_NULL, N_CLASSES };  // C_LIT: true / false
constexpr int SCRIPT_END = -1;
constexpr int JMP_BASE = -20000, CHOICE_BASE = -30000, MAX_ALTS = 16;
enum { K_LIT, K_END, K_SLOT, K_JMP, K_CHOICE };
__device__ inline int entry_kind(int e) { return e >= 0 ? K_LIT : e == SCRIPT_END ? K_END : e > JMP_BASE ? K_SLOT : e > CHOICE_BASE ? K_JMP : K_CHOICE; }
// the cursor with jumps followed (compiled scripts chain two at most)
__device__ inline int resolve(const int* __restrict__ ent, int cur, int L) {
  for (int hop = 0; hop < 8; ++hop) {
    if (cur < 0 || cur >= L || entry_kind(ent[cur]) != K_JMP) break;
    cur = JMP_BASE - ent[cur];
  }
  return cur;
}
enum { S_OFF, S_START, S_OBJ_FIRST, S_KEY, S_AFTER_KEY, S_VALUE, S_STR, S_NUM, S_AFTER_VALUE, S_OBJ_NEXT, S_ARR_FIRST, S_DONE, S_END };
constexpr int MAX_DEPTH = 16;
constexpr int TPB = 1024;

struct St { int s, depth, stack; };

__device__ inline St pop(St a) {
  a.depth -= 1;
  a.stack &= ~(1 << a.depth);
  a.s = a.depth == 0 ? S_DONE : S_AFTER_VALUE;
  return a;
}

// the automaton's move on a token of class c; ok = false when the grammar forbids it
__device__ inline St transition(St a, int c, bool& ok) {
  ok = true;
  const bool top_arr = a.depth > 0 && ((a.stack >> (a.depth - 1)) & 1);
  const bool in_string = c != C_OTHER && c != C_QUOTE && c != C_EOS;
  St n = a;
  switch (a.s) {
    case S_START:
      if (c == C_LBRACE) { n.s = S_OBJ_FIRST; n.depth = 1; n.stack = 0; return n; }
      break;
    case S_OBJ_FIRST:
      if (c == C_QUOTE) { n.s = S_KEY; return n; }
      if (c == C_RBRACE) return pop(a);
      break;
    case S_KEY:
      if (c == C_QUOTE) { n.s = S_AFTER_KEY; return n; }
      if (in_string) return n;
      break;
    case S_AFTER_KEY:
      if (c == C_COLON) { n.s = S_VALUE; return n; }
      break;
    case S_VALUE:
    case S_ARR_FIRST:
      if (c == C_QUOTE) { n.s = S_STR; return n; }
      if (c == C_LBRACE && a.depth < MAX_DEPTH) { n.s = S_OBJ_FIRST; n.depth = a.depth + 1; return n; }
      if (c == C_LBRACK && a.depth < MAX_DEPTH) { n.s = S_ARR_FIRST; n.depth = a.depth + 1; n.stack = a.stack | (1 << a.depth); return n; }
      if (c == C_DIGIT_NZ) { n.s = S_NUM; return n; }
      if (c == C_LIT || c == C_NULL) { n.s = S_AFTER_VALUE; return n; }
      if (a.s == S_ARR_FIRST && c == C_RBRACK) return pop(a);
      break;
    case S_STR:
      if (c == C_QUOTE) { n.s = S_AFTER_VALUE; return n; }
      if (in_string) return n;
      break;
    case S_NUM:
    case S_AFTER_VALUE:
      if (a.s == S_NUM && (c == C_DIGIT || c == C_DIGIT_NZ)) return n;
      if (c == C_COMMA) { n.s = top_arr ? S_VALUE : S_OBJ_NEXT; return n; }
      if ((c == C_RBRACK && top_arr) || (c == C_RBRACE && !top_arr)) return pop(a);
      break;
    case S_OBJ_NEXT:
      if (c == C_QUOTE) { n.s = S_KEY; return n; }
      break;
    case S_DONE:
      if (c == C_EOS) { n.s = S_END; n.depth = 0; n.stack = 0; return n; }
      break;
    default:
      break;
  }
  ok = false;
  return a;
}

// fewest tokens from a state to the end of the sequence (closers + EOS); vc = cost of a value
__device__ inline int close_cost(int s, int depth, int vc) {
  const int tail = depth + 1;
  switch (s) {
    case S_START: return 3;
    case S_KEY: return 2 + vc + tail;
    case S_AFTER_KEY: return 1 + vc + tail;
    case S_VALUE: return vc + tail;
    case S_STR: return 1 + tail;
    case S_OBJ_NEXT: return 3 + vc + tail;
    case S_OBJ_FIRST:
    case S_ARR_FIRST:
    case S_NUM:
    case S_AFTER_VALUE: return tail;
    case S_DONE: return 1;
    default: return 0;
  }
}

// Bits 16.. of the stack word restrict the NEXT token to a set of classes (0 = none); the first
// advance clears them.  The restriction is dropped if the grammar allows none of its classes.
constexpr int FIRST_SHIFT = 16;

__device__ inline unsigned allowed_classes(St a, int left, int vc) {
  if (a.s == S_OFF || a.s == S_END) return (1u << N_CLASSES) - 1u;
  const unsigned first = (unsigned)a.stack >> FIRST_SHIFT;
  a.stack &= (1 << FIRST_SHIFT) - 1;
  unsigned fits = 0, grammar = 0;
#pragma unroll
  for (int c = 1; c < N_CLASSES; ++c) {
    bool ok;
    const St n = transition(a, c, ok);
    if (!ok) continue;
    grammar |= 1u << c;
    if (close_cost(n.s, n.depth, vc) <= left - 1) fits |= 1u << c;
  }
  if (first & grammar) {
    grammar &= first;
    fits &= first;
  }
  return fits ? fits : grammar;
}

// a value has just been completed at the top level (or a literal is being written): the
// script's next literal token comes next
__device__ inline bool forcing(int s, int depth) { return depth == 1 && (s == S_AFTER_VALUE || s == S_NUM || s == S_OBJ_NEXT); }

// (fits, grammar) class masks of the FIRST token of a value slot at the top level
__device__ inline void slot_classes(unsigned mask, int after, int left, int vc, unsigned& fits, unsigned& grammar) {
  fits = 0; grammar = 0;
  const St v = {S_VALUE, 1, 0};
#pragma unroll
  for (int c = 1; c < N_CLASSES; ++c) {
    bool ok;
    const St n = transition(v, c, ok);
    if (!ok || n.depth < 1) continue;
    grammar |= 1u << c;
    if (close_cost(n.s, n.depth, vc) - 2 + after <= left - 1) fits |= 1u << c;
  }
  if (mask & grammar) { grammar &= mask; fits &= mask; }
}

// scripted sequences: class mask + up to MAX_ALTS tokens that are allowed whatever their class
__device__ inline unsigned allowed_scripted(St a, int left, int vc, const int* __restrict__ ent, const int* __restrict__ rest, int cur, int L, int* toks, int& n_toks) {
  n_toks = 0;
  if (ent == nullptr || cur < 0 || cur >= L || ent[cur] == SCRIPT_END || a.s == S_OFF || a.s == S_END) return allowed_classes(a, left, vc);
  const int e = ent[cur];
  const int kind = entry_kind(e);
  a.stack &= (1 << FIRST_SHIFT) - 1;
  if (kind != K_SLOT && forcing(a.s, a.depth)) {
    const bool more_digits = a.s == S_NUM && left - 1 >= rest[cur];
    const unsigned digits = more_digits ? ((1u << C_DIGIT) | (1u << C_DIGIT_NZ)) : 0u;
    if (kind == K_LIT) {
      toks[0] = e;
      n_toks = 1;
      return digits;
    }
    // a choice: every alternative after which the document can still be closed in the tokens left
    int n = CHOICE_BASE - e;
    n = n > MAX_ALTS ? MAX_ALTS : n;
    unsigned cls_fit = 0, cls_all = 0;
    int n_fit = 0;
    for (int k = 0; k < n; ++k) {
      const int key = ent[cur + 1 + 2 * k], tgt = ent[cur + 2 + 2 * k];
      const int after = (tgt >= 0 && tgt < L) ? rest[tgt] : 1;
      if (key >= 0) {
        if (after <= left - 1) toks[n_fit++] = key;
      } else {
        unsigned f, g;
        slot_classes((unsigned)(-2 - key), after, left, vc, f, g);
        cls_fit |= f;
        cls_all |= g;
      }
    }
    if (n_fit > 0 || cls_fit != 0) {
      n_toks = n_fit;
      return cls_fit | digits;
    }
    for (int k = 0; k < n; ++k) {  // budget too small from the start: stay in the script, the output is cut short
      const int key = ent[cur + 1 + 2 * k];
      if (key >= 0) toks[n_toks++] = key;
    }
    return cls_all | digits;
  }
  const int after = kind == K_SLOT ? (cur + 1 < L ? rest[cur + 1] : 1) : rest[cur];
  const unsigned first = (kind == K_SLOT && a.s == S_VALUE && a.depth == 1) ? (unsigned)(-2 - e) : 0u;
  unsigned fits = 0, grammar = 0;
#pragma unroll
  for (int c = 1; c < N_CLASSES; ++c) {
    bool ok;
    const St n = transition(a, c, ok);
    if (!ok || n.depth < 1) continue;  // closing the top-level object is the script's business
    if (a.depth == 1 && (a.s == S_AFTER_VALUE || a.s == S_NUM) && (c == C_COMMA || c == C_RBRACE)) continue;
    grammar |= 1u << c;
    if (close_cost(n.s, n.depth, vc) - 2 + after <= left - 1) fits |= 1u << c;
  }
  if (first & grammar) {
    grammar &= first;
    fits &= first;
  }
  return fits ? fits : grammar;
}

typedef __attribute__((ext_vector_type(8))) unsigned short us8;

__global__ __launch_bounds__(TPB) void guided_mask_kernel(unsigned short* __restrict__ logits, long long row_stride, int V, int n_cls, const long long* __restrict__ rows,
                                                           const int* __restrict__ slots, const int* __restrict__ init, int* __restrict__ state,
                                                           const unsigned char* __restrict__ cls, int vc, const int* __restrict__ script,
                                                           int* __restrict__ cursor, const int* __restrict__ cur_init, int L) {
  const int r = blockIdx.x, tid = threadIdx.x;
  const int slot = slots[r];
  const bool fresh = init[4 * r] >= 0;
  const int* src = fresh ? init + 4 * r : state + 4 * slot;
  const St a = {src[0], src[1], src[2]};
  const int left = src[3];
  int cur = -1;
  if (script != nullptr) cur = fresh ? cur_init[r] : cursor[slot];
  if (fresh && tid < 4) state[4 * slot + tid] = init[4 * r + tid];  // nobody reads state[] / cursor[] of this slot in this launch
  if (fresh && tid == 4 && script != nullptr) cursor[slot] = cur;
  // the allowed set is worked out once per row and shared (it walks a choice's alternatives)
  __shared__ unsigned allow_s;
  __shared__ int toks_s[MAX_ALTS], n_toks_s;
  if (tid == 0) {
    const int* ent0 = script != nullptr ? script + (long long)slot * 2 * L : nullptr;
    int tk[MAX_ALTS], nt = 0;
    allow_s = allowed_scripted(a, left, vc, ent0, ent0 != nullptr ? ent0 + L : nullptr, cur, L, tk, nt);
    n_toks_s = nt;
    for (int k = 0; k < nt; ++k) toks_s[k] = tk[k];
  }
  __syncthreads();
  const unsigned allow = allow_s;
  const int n_toks = n_toks_s;
  if (allow == (1u << N_CLASSES) - 1u) return;
  // Tokens that are allowed whatever their class (the forced literal, the keys of a choice) keep
  // their logits: thread k saves the k-th one before the row is masked by class and puts it back
  // afterwards (all barriers below are taken by every thread: n_toks is the same for all of them).
  unsigned short* row = logits + rows[r] * row_stride;
  unsigned short saved = 0;
  int mine = -1;
  if (tid < n_toks) {
    mine = toks_s[tid];
    if (mine >= 0 && mine < V) saved = row[mine]; else mine = -1;
  }
  __syncthreads();
  constexpr unsigned short NEG_INF = 0xFF80;
  int done = 0;
  if ((((uintptr_t)row) & 15) == 0 && (((uintptr_t)cls) & 7) == 0) {
    const int nv = n_cls >> 3;
    for (int i = tid; i < nv; i += TPB) {
      us8 x = reinterpret_cast<const us8*>(row)[i];
      const unsigned long long c8 = reinterpret_cast<const unsigned long long*>(cls)[i];
      bool any = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned c = (unsigned)(c8 >> (8 * j)) & 0xffu;
        if (!((allow >> c) & 1u)) { x[j] = NEG_INF; any = true; }
      }
      if (any) reinterpret_cast<us8*>(row)[i] = x;
    }
    done = nv * 8;
  }
  for (int i = done + tid; i < V; i += TPB) {
    const unsigned c = i < n_cls ? cls[i] : (unsigned)C_OTHER;  // ids past the tokenizer's vocabulary (padding columns) are never allowed
    if (!((allow >> c) & 1u)) row[i] = NEG_INF;
  }
  if (n_toks > 0) {
    __syncthreads();  // every masking store of this workgroup is complete (and visible to it) before the listed tokens are restored
    if (mine >= 0) row[mine] = saved;
  }
}

__global__ void guided_advance_kernel(const long long* __restrict__ out_ids, const long long* __restrict__ rows, const int* __restrict__ slots, int R,
                                      int* __restrict__ state, const unsigned char* __restrict__ cls, int n_cls, const int* __restrict__ script,
                                      int* __restrict__ cursor, int L) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int slot = slots[r];
  int* st = state + 4 * slot;
  St a = {st[0], st[1], st[2] & ((1 << FIRST_SHIFT) - 1)};
  if (a.s == S_OFF || a.s == S_END) return;
  const long long tok = out_ids[rows[r]];
  const int c = (tok >= 0 && tok < n_cls) ? cls[tok] : (int)C_OTHER;
  if (script != nullptr) {
    const int cur = cursor[slot];
    const int* ent = script + (long long)slot * 2 * L;
    if (cur >= 0 && cur < L && ent[cur] != SCRIPT_END) {
      const int e = ent[cur];
      const int kind = entry_kind(e);
      // a skeleton token was written (a literal, or the token that picked an alternative): the cursor
      // moves and the automaton is set by what the script holds there
      auto land = [&](int at) {
        const int nx = resolve(ent, at, L);
        const int k2 = (nx >= 0 && nx < L) ? entry_kind(ent[nx]) : (int)K_END;
        cursor[slot] = nx;
        st[0] = k2 == K_END ? S_DONE : (k2 == K_SLOT ? S_VALUE : S_OBJ_NEXT);
        st[1] = k2 == K_END ? 0 : 1;
        st[2] = 0;
        st[3] = st[3] - 1;
      };
      if (kind == K_LIT && forcing(a.s, a.depth) && tok == (long long)e) {
        land(cur + 1);
        return;
      }
      if (kind == K_CHOICE && forcing(a.s, a.depth)) {
        int n = CHOICE_BASE - e;
        n = n > MAX_ALTS ? MAX_ALTS : n;
        for (int k = 0; k < n; ++k) {
          const int key = ent[cur + 1 + 2 * k];
          if (key >= 0 && tok == (long long)key) {
            land(ent[cur + 2 + 2 * k]);
            return;
          }
        }
        if (!(a.s == S_NUM && (c == C_DIGIT || c == C_DIGIT_NZ))) {  // (a digit goes on with the number in front of the choice)
          for (int k = 0; k < n; ++k) {
            const int key = ent[cur + 1 + 2 * k];
            if (key >= 0) continue;
            const unsigned mask = (unsigned)(-2 - key);
            bool ok2;
            const St v = {S_VALUE, 1, 0};
            const St n2 = transition(v, c, ok2);
            if (ok2 && n2.depth >= 1 && (mask == 0u || ((mask >> c) & 1u))) {
              cursor[slot] = resolve(ent, ent[cur + 2 + 2 * k], L);
              st[0] = n2.s; st[1] = n2.depth; st[2] = n2.stack; st[3] = st[3] - 1;
              return;
            }
          }
        }
      }
      if (kind == K_SLOT && a.s == S_VALUE && a.depth == 1) cursor[slot] = resolve(ent, cur + 1, L);  // the slot's value has begun
    }
  }
  bool ok;
  const St n = transition(a, c, ok);
  st[0] = n.s; st[1] = n.depth; st[2] = n.stack; st[3] = st[3] - 1;
}

}  // namespace

// script: int32 [slots][2][L] (entries, rest) or nullptr (no sequence is scripted); cursor: int32 [slots];
// cur_init: int32 [R], the cursor a row's slot starts from when init (re)sets it (-1: no script)
extern "C" int kv_guided_mask(void* logits, long long row_stride, int V, int n_cls, const long long* rows, const int* slots, const int* init, int R, int* state,
                              const unsigned char* cls, int vc, const int* script, int* cursor, const int* cur_init, int L, hipStream_t stream) {
  if (R <= 0) return 0;
  if (V <= 0 || n_cls > V) return -1;
  if (script != nullptr && (cursor == nullptr || cur_init == nullptr || L < 1)) return -1;
  hipLaunchKernelGGL(guided_mask_kernel, dim3(R), dim3(TPB), 0, stream, (unsigned short*)logits, row_stride, V, n_cls, rows, slots, init, state, cls, vc, script, cursor,
                     cur_init, L);
  return (int)hipGetLastError();
}

extern "C" int kv_guided_advance(const long long* out_ids, const long long* rows, const int* slots, int R, int* state, const unsigned char* cls, int n_cls,
                                 const int* script, int* cursor, int L, hipStream_t stream) {
  if (R <= 0) return 0;
  if (script != nullptr && (cursor == nullptr || L < 1)) return -1;
  hipLaunchKernelGGL(guided_advance_kernel, dim3((R + 63) / 64), dim3(64), 0, stream, out_ids, rows, slots, R, state, cls, n_cls, script, cursor, L);
  return (int)hipGetLastError();
}
