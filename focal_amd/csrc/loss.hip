// FOCAL loss head (models/loss.py:139-218) forward + backward in one call.
//
// The reference materialises [seq, 2b, 2b, d] cosine broadcasts and index-masks them; here every cross-sample term
// is a pair of batched exact-fp32 MFMA products (similarity / Gram matrix, then gradient = coefficient matrix x
// embeddings) with small row-wise kernels in between (wave-shuffle log-sum-exp, block means, hinges):
//   InfoNCE   S = Zn Zn^T / T          -> lse rows, loss   -> W = softmax + softmax^T - 2*pos   -> dZn = W Zn
//   ranking   G = X X^T -> D = cdist   -> block means, hinge -> E = (A + A^T) / D              -> dX = rowsum(E) X - E X
//   orthogonality is row-local (one wave per sample).
// Everything is fp32: with T = 0.07 the logits amplify cosine errors 14x.
//
// Row sharding over data-parallel ranks (focal_loss_head_shard_a / _b).  Every cross-sample matrix here is symmetric and the
// gradient of a sample needs only ITS row of it: rank r of W owns the b / W subsequences it contributed (rows r0 .. r0 + bl - 1 of
// each view half) and evaluates, over ALL columns, only those rows of S, of the softmax coefficients, of the distance matrix and of
// its coefficients -- 1 / W of every GEMM and row pass.  Two things of the other ranks' rows enter a row of coefficients: the
// log-sum-exp of the column's own row (softmax^T term) and the diagonal block mean of the column's subsequence (the transposed hinge).
// Both are tiny vectors; phase A writes this rank's part of them (and its partial loss terms) into one exchange chunk, the caller
// all-gathers the chunks (ONE small collective), phase B finishes.  focal_loss_head is the same code with one rank and no collective.
struct Shard {
  int r0, bl;        // first own subsequence, own subsequences (b / world)
  int world, ch;     // ranks, floats per exchange chunk
  int o_diag, o_terms;  // offsets inside a chunk: [lse of own rows: nblk_total * 2 bl | diag means: Q * bl | partial terms: 5 | pad]
};
#include <stdlib.h>
#include "gemm.hpp"

#define LOSS_MAXP 20   // InfoNCE problems (2 views x mod pairs + mods = M^2 <= 16) and orthogonality problems (M (M + 1) <= 20) share the table type
#define LOSS_MAXQ 8    // ranking problems  (2 views x mods)
#define LOSS_MAXO 20   // orthogonality problems

struct PairProb { const float* e1; const float* e2; float* d1; float* d2; int off1, off2, kind; };
struct PairTable { PairProb p[LOSS_MAXP]; };
struct RankTable { const float* x[LOSS_MAXQ]; float* d[LOSS_MAXQ]; };

// Loss terms are sums over thousands of rows.  One atomic per row-wave onto the same word is a serial chain (~10 ns a link:
// 8 k rows = 0.1 ms at a global batch of 2048, more than the similarity GEMM); the row kernels therefore run as <= LOSS_ROW_BLOCKS
// persistent 4-wave workgroups, keep their contribution in a register across rows and issue ONE atomic per workgroup and term.
#define LOSS_ROW_BLOCKS 512
__device__ __forceinline__ void block_term_add(float v, float* dst) {  // v: per-wave value (lane 0 holds it); 256-thread workgroup
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) part[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = part[0] + part[1] + part[2] + part[3];
    if (t != 0.f) atomicAdd(dst, t);
  }
}
static inline int loss_row_blocks(long rows) {
  const long b = (rows + 3) / 4;
  return (int)(b > LOSS_ROW_BLOCKS ? LOSS_ROW_BLOCKS : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------------------------------------ InfoNCE
// row r of problem (p, t): r < b -> e1[(r)*seq + t], else e2[(r-b)*seq + t]; transposition of loss.py:64-73.
// Every (problem, step) block holds n2p = round_up(2b, 4) rows: the fp32 GEMM moves 16-byte chunks along its reduction index, so
// the logical 2b (any b >= 2: the last batch of an epoch may hold an odd number of subsequences) is padded with zero rows /
// columns that every row kernel skips.
__device__ __forceinline__ void nce_pack_body(const int bid, const int nb, const PairTable& tab, int p0, int nprob, int seq, int b, int n2p, int dim, int width,
                                                       float* __restrict__ Zn, float* __restrict__ nrm, float* __restrict__ zero5) {
  const int lane = threadIdx.x & 63;
  if (zero5 != nullptr && bid == 0 && threadIdx.x < 5) zero5[threadIdx.x] = 0.f;  // this rank's partial loss terms (first launch of the head)
  const long row = (long)bid * 4 + (threadIdx.x >> 6);
  const long rows = (long)nprob * seq * n2p;
  if (row >= rows) return;
  const int r = row % n2p, t = (row / n2p) % seq, p = row / ((long)n2p * seq);
  if (r >= 2 * b) {  // padding row
    for (int c = lane; c < width; c += 64) Zn[row * width + c] = 0.f;
    if (lane == 0) nrm[row] = 1.f;
    return;
  }
  const PairProb pr = tab.p[p0 + p];
  const float* src = (r < b) ? pr.e1 + ((long)r * seq + t) * dim + pr.off1 : pr.e2 + ((long)(r - b) * seq + t) * dim + pr.off2;
  float ss = 0.f;
  for (int c = lane; c < width; c += 64) { const float v = src[c]; ss += v * v; }
  ss = wave_sum(ss);
  const float n = fmaxf(sqrtf(ss), 1e-8f);  // nn.CosineSimilarity eps, loss.py:15
  for (int c = lane; c < width; c += 64) Zn[row * width + c] = src[c] / n;
  if (lane == 0) nrm[row] = n;
}

// one wave per OWN row i of S[p,t]: lse over j != i, loss_i = lse_i - S[i][pos(i)].  Own row w of block blk (w = h * bl + k: view half
// h, own subsequence k) is row i = h * b + r0 + k of the block; its lse goes to slot (gb0 + blk) * 2 bl + w of this rank's chunk.
__device__ __forceinline__ void nce_rows_body(const int bid, const int nb, const PairTable& tab, int p0, int nprob, int seq, int b, int n2p, Shard sh, int gb0,
                                                       const float* __restrict__ S, float* __restrict__ lse_own, float* __restrict__ lse_full,
                                                       float* __restrict__ terms) {
  const int lane = threadIdx.x & 63;
  const int n2 = 2 * b, own = 2 * sh.bl;
  const long rows = (long)nprob * seq * own;
  float acc0 = 0.f, acc1 = 0.f;  // contributions to terms[0] (shared family) / terms[1] (private family)
  for (long idx = (long)bid * 4 + (threadIdx.x >> 6); idx < rows; idx += (long)nb * 4) {
    const long blk = idx / own;
    const int w = idx % own, i = (w / sh.bl) * b + sh.r0 + (w % sh.bl);
    const int p = blk / seq;
    const int kind = tab.p[p0 + p].kind;
    const float* s = S + (blk * n2p + i) * n2p;
    float mx = -3.0e38f;
    for (int j = lane; j < n2; j += 64) if (j != i) mx = fmaxf(mx, s[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n2; j += 64) if (j != i) sum += __expf(s[j] - mx);
    sum = wave_sum(sum);
    const float l = mx + __logf(sum);
    if (lane == 0) {
      lse_own[(gb0 + blk) * own + w] = l;
      lse_full[blk * n2p + i] = l;
      const float contrib = (l - s[(i + b) % n2]) / (float)(seq * n2);
      if (kind == 0) acc0 += contrib; else acc1 += contrib;
    }
  }
  block_term_add(acc0, terms + 0);
  block_term_add(acc1, terms + 1);
}

// The gathered chunks [rank][lse of its rows | its diagonal means | ...] -> lse in block-row order [gb][n2p] and the diagonal means
// [q][b] (so that the element-wise kernels index them without a division per element)
__global__ __launch_bounds__(256) void xchg_unpack_kernel(Shard sh, int nblk, int b, int n2p, int Q, const float* __restrict__ xall,
                                                          float* __restrict__ lse, float* __restrict__ diag) {
  const int n2 = 2 * b;
  const long nl = (long)nblk * n2, total = nl + (long)Q * b;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    if (e < nl) {
      const long gb = e / n2;
      const int j = e % n2, h = j >= b, s = j - h * b;
      lse[gb * n2p + j] = xall[(long)(s / sh.bl) * sh.ch + gb * 2 * sh.bl + h * sh.bl + (s % sh.bl)];
    } else {
      const long t = e - nl;
      const int q = t / b, J = t % b;
      diag[t] = xall[(long)(J / sh.bl) * sh.ch + sh.o_diag + q * sh.bl + (J % sh.bl)];
    }
  }
}

// in place, OWN rows of S -> W (times the family weight): W_ij = [j != i](e^{S_ij - lse_i} + e^{S_ij - lse_j}) - 2 [j == pos(i)]
__device__ __forceinline__ void nce_weights_body(const int bid, const int nb, const PairTable& tab, int p0, int nprob, int seq, int b, int n2p, Shard sh,
                                                          float* __restrict__ S, const float* __restrict__ lse, float w_shared, float w_private) {
  const int n2 = 2 * b, own = 2 * sh.bl;
  const long total = (long)nprob * seq * own * n2p;
  for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
    const int j = e % n2p;
    const long idx = e / n2p, blk = idx / own;
    const int w = idx % own, i = (w / sh.bl) * b + sh.r0 + (w % sh.bl);
    const int p = blk / seq;
    const float wk = tab.p[p0 + p].kind == 0 ? w_shared : w_private;
    float* sp = S + (blk * n2p + i) * n2p + j;
    float wv = 0.f;
    if (j != i && j < n2) {
      const float sv = *sp;
      wv = __expf(sv - lse[blk * n2p + i]) + __expf(sv - lse[blk * n2p + j]);
      if (j == (i + b) % n2) wv -= 2.0f;
    }
    *sp = wv * wk;
  }
}

// dz = (dzn - zn (zn . dzn)) / ||z||, scattered (+=) to the right sample / half of the source embeddings; own rows only
__device__ __forceinline__ void nce_unpack_body(const int bid, const int nb, const PairTable& tab, int p0, int nprob, int seq, int b, int n2p, int dim, int width, Shard sh,
                                                         const float* __restrict__ Zn, const float* __restrict__ nrm,
                                                         const float* __restrict__ dZn) {
  const int lane = threadIdx.x & 63, own = 2 * sh.bl;
  const long idx = (long)bid * 4 + (threadIdx.x >> 6);
  if (idx >= (long)nprob * seq * own) return;
  const long blk = idx / own;
  const int w = idx % own, h = w / sh.bl, sq_ = sh.r0 + (w % sh.bl);  // view half, subsequence
  const int t = blk % seq, p = blk / seq;
  const long row = blk * n2p + h * b + sq_;
  const PairProb pr = tab.p[p0 + p];
  float* dst = (h == 0) ? pr.d1 + ((long)sq_ * seq + t) * dim + pr.off1 : pr.d2 + ((long)sq_ * seq + t) * dim + pr.off2;
  float dot = 0.f;
  for (int c = lane; c < width; c += 64) dot += Zn[row * width + c] * dZn[row * width + c];
  dot = wave_sum(dot);
  const float inv = 1.0f / nrm[row];
  for (int c = lane; c < width; c += 64) atomicAdd(dst + c, (dZn[row * width + c] - Zn[row * width + c] * dot) * inv);
}

// ------------------------------------------------------------------------------------------------ ranking
// (Bp = round_up(B, 4) rows per problem, zero padding: same reason as n2p above)
__device__ __forceinline__ void rank_pack_body(const int bid, const int nb, const RankTable& tab, int nq, int B, int Bp, int dim, float* __restrict__ X,
                                                        float* __restrict__ sq) {
  const int lane = threadIdx.x & 63;
  const long row = (long)bid * 4 + (threadIdx.x >> 6);
  if (row >= (long)nq * Bp) return;
  const int q = row / Bp, s = row % Bp;
  if (s >= B) {
    for (int c = lane; c < dim; c += 64) X[row * dim + c] = 0.f;
    if (lane == 0) sq[row] = 0.f;
    return;
  }
  const float* src = tab.x[q] + (long)s * dim;
  float ss = 0.f;
  for (int c = lane; c < dim; c += 64) { const float v = src[c]; X[row * dim + c] = v; ss += v * v; }
  ss = wave_sum(ss);
  if (lane == 0) sq[row] = ss;
}

// in place, own rows [rs0, rs0 + nr) of G -> D = sqrt(max(0, |x_p|^2 + |x_q|^2 - 2 G_pq)), D_pp = 0   (torch.cdist mm path, loss.py:117)
__device__ __forceinline__ void rank_dist_body(const int bid, const int nb, int nq, int B, int Bp, int rs0, int nr, float* __restrict__ G, const float* __restrict__ sq) {
  const long total = (long)nq * nr * Bp;
  for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
    const int c = e % Bp;
    const int r = rs0 + (int)((e / Bp) % nr);
    const long qb = (e / ((long)Bp * nr)) * Bp;
    const long at = (qb + r) * Bp + c;
    const float d2 = sq[qb + r] + sq[qb + c] - 2.0f * G[at];
    G[at] = (r == c || c >= B) ? 0.f : sqrtf(fmaxf(d2, 0.f));
  }
}

// Dbar[q][I][J] = mean of the seq x seq block (self pairs excluded), loss.py:118-124; own subsequences I, all J
__device__ __forceinline__ void rank_blockmean_body(const int bid, const int nb, int nq, int b, int seq, int Bp, Shard sh, const float* __restrict__ D, float* __restrict__ Dbar) {
  const long total = (long)nq * sh.bl * b;
  for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
    const int J = e % b, I = sh.r0 + (int)((e / b) % sh.bl), q = e / ((long)b * sh.bl);
    const float* base = D + ((long)q * Bp + (long)I * seq) * Bp + (long)J * seq;
    float s = 0.f;
    for (int a = 0; a < seq; ++a)
      for (int c = 0; c < seq; ++c) s += base[(long)a * Bp + c];  // the self pairs hold exact zeros
    Dbar[((long)q * b + I) * b + J] = s / (float)(seq * seq - (I == J ? seq : 0));
  }
}

// one wave per (q, own I): hinge over J != I (MarginRankingLoss(margin, y = -1), loss.py:127-135) and dL/dDbar of that row; the
// row's diagonal mean goes to the exchange chunk (other ranks need it for the transposed hinge of their coefficient rows)
__device__ __forceinline__ void rank_hinge_body(const int bid, const int nb, int nq, int b, float margin, Shard sh, const float* __restrict__ Dbar,
                                                         float* __restrict__ dDbar, float* __restrict__ diag_own, float* __restrict__ diag_full,
                                                         float* __restrict__ terms) {
  const int lane = threadIdx.x & 63;
  const float inv = 1.0f / ((float)b * (float)(b - 1));
  float acc = 0.f;
  for (long idx = (long)bid * 4 + (threadIdx.x >> 6); idx < (long)nq * sh.bl; idx += (long)nb * 4) {
    const int I = sh.r0 + (int)(idx % sh.bl);
    const long row = (idx / sh.bl) * b + I;
    const float* d = Dbar + row * b;
    const float dii = d[I];
    float loss = 0.f, cnt = 0.f;
    for (int J = lane; J < b; J += 64) {
      if (J == I) continue;
      const float h = dii - d[J] + margin;
      const bool on = h > 0.f;
      loss += on ? h : 0.f;
      cnt += on ? 1.f : 0.f;
      dDbar[row * b + J] = on ? -inv : 0.f;
    }
    loss = wave_sum(loss);
    cnt = wave_sum(cnt);
    if (lane == 0) {
      dDbar[row * b + I] = cnt * inv;
      diag_own[idx] = dii;
      diag_full[row] = dii;
      acc += loss * inv;
    }
  }
  block_term_add(acc, terms + 3);
}

// in place, own rows of D -> E = w_rank * (A_pq + A_qp) / D_pq, A_pq = dDbar[I(p)][J(q)] / count(I, J); rowsum[p] = sum_q E_pq.
// dDbar[J][I] of a column's subsequence J is re-derived from its diagonal mean (gathered: diag[q][J]) and Dbar[I][J] (block means are symmetric
// up to the summation order of the 16 distances): -inv where the hinge of row J is active against I.
__device__ __forceinline__ void rank_coeff_body(const int bid, const int nb, int nq, int b, int seq, int Bp, float w_rank, float margin, Shard sh, float* __restrict__ D,
                                                         const float* __restrict__ Dbar, const float* __restrict__ dDbar,
                                                         const float* __restrict__ diag, float* __restrict__ rowsum) {
  const int lane = threadIdx.x & 63;
  const int B = b * seq, nr = sh.bl * seq;
  const float inv = 1.0f / ((float)b * (float)(b - 1));
  const long idx = (long)bid * 4 + (threadIdx.x >> 6);
  if (idx >= (long)nq * nr) return;
  const int q = idx / nr, p = sh.r0 * seq + (int)(idx % nr), I = p / seq;
  const long row = (long)q * Bp + p;
  const float* dd = dDbar + ((long)q * b + I) * b;
  const float* db = Dbar + ((long)q * b + I) * b;
  float* drow = D + row * Bp;
  float rs = 0.f;
  for (int c = lane; c < Bp; c += 64) {
    float e = 0.f;
    if (c != p && c < B) {
      const int J = c / seq;
      const float dist = drow[c];
      if (dist > 1e-12f) {
        const float cnt = (float)(seq * seq - (I == J ? seq : 0));
        float a = dd[J];
        if (J == I) a += a;
        else if (diag[q * b + J] - db[J] + margin > 0.f) a -= inv;
        e = w_rank * a / (cnt * dist);
      }
    }
    drow[c] = e;
    rs += e;
  }
  rs = wave_sum(rs);
  if (lane == 0) rowsum[row] = rs;
}

__device__ __forceinline__ void rank_grad_body(const int bid, const int nb, const RankTable& tab, int nq, int Bp, int dim, int rs0, int nr, const float* __restrict__ X,
                                                        const float* __restrict__ rowsum, const float* __restrict__ EX) {
  const long total = (long)nq * nr * dim;
  for (long e = (long)bid * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
    const int c = e % dim, s = rs0 + (int)((e / dim) % nr), q = e / ((long)dim * nr);
    const long at = ((long)q * Bp + s) * dim + c;
    atomicAdd(tab.d[q] + (long)s * dim + c, rowsum[(long)q * Bp + s] * X[at] - EX[at]);
  }
}

// ------------------------------------------------------------------------------------------------ orthogonality
// CosineEmbeddingLoss(target = -1, margin 0, mean) = mean(max(0, cos)), loss.py:89-106; one wave per sample.
__device__ __forceinline__ void orth_body(const int bid, const int nb, const PairTable& tab, int nprob, int B, int dim, int width, float w_orth, int rs0, int nr,
                                                   float* __restrict__ terms) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (long idx = (long)bid * 4 + (threadIdx.x >> 6); idx < (long)nprob * nr; idx += (long)nb * 4) {
    const int p = idx / nr, s = rs0 + (int)(idx % nr);
    const PairProb pr = tab.p[p];
    const float* x1 = pr.e1 + (long)s * dim + pr.off1;
    const float* x2 = pr.e2 + (long)s * dim + pr.off2;
    float d = 0.f, n1 = 0.f, n2 = 0.f;
    for (int c = lane; c < width; c += 64) { const float a = x1[c], b2 = x2[c]; d += a * b2; n1 += a * a; n2 += b2 * b2; }
    d = wave_sum(d); n1 = wave_sum(n1) + 1e-12f; n2 = wave_sum(n2) + 1e-12f;
    const float den = sqrtf(n1 * n2);
    const float cs = d / den;
    if (cs > 0.f) {
      const float k = w_orth / (float)B;
      float* g1 = pr.d1 + (long)s * dim + pr.off1;
      float* g2 = pr.d2 + (long)s * dim + pr.off2;
      for (int c = lane; c < width; c += 64) {
        const float a = x1[c], b2 = x2[c];
        atomicAdd(g1 + c, k * (b2 / den - cs * a / n1));
        atomicAdd(g2 + c, k * (a / den - cs * b2 / n2));
      }
      if (lane == 0) acc += cs / (float)B;
    }
  }
  block_term_add(acc, terms + 2);
}

// rank_dist + rank_blockmean + rank_hinge of one (problem q, own subsequence I) in ONE pass, one wave per row: lane J turns the seq x seq
// block (I, J) of the Gram matrix into distances (written back in place: phase B reads them), keeps the block mean, and the hinge over
// the row's b means follows through a wave-private LDS row (b <= RANK_ROW_MAX_B; longer rows use the three separate passes).
#define RANK_ROW_MAX_B 1024
__device__ __forceinline__ void rank_rows_body(const int bid, const int nb, int nq, int b, int seq, int B, int Bp, float margin, Shard sh,
                                               float* __restrict__ G, const float* __restrict__ sq, float* __restrict__ Dbar,
                                               float* __restrict__ dDbar, float* __restrict__ diag_own, float* __restrict__ diag_full,
                                               float* __restrict__ terms) {
  __shared__ float means[4][RANK_ROW_MAX_B];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* mrow = means[wave];
  const float inv = 1.0f / ((float)b * (float)(b - 1));
  float acc = 0.f;
  for (long idx = (long)bid * 4 + wave; idx < (long)nq * sh.bl; idx += (long)nb * 4) {
    const int I = sh.r0 + (int)(idx % sh.bl), q = (int)(idx / sh.bl);
    const long qb = (long)q * Bp, row = (long)q * b + I;
    for (int J = lane; J < b; J += 64) {
      float s = 0.f;
      for (int a = 0; a < seq; ++a) {
        const int p = I * seq + a;
        const float sp = sq[qb + p];
        float* g = G + (qb + p) * Bp + (long)J * seq;
        for (int c = 0; c < seq; ++c) {
          const int col = J * seq + c;
          const float d2 = sp + sq[qb + col] - 2.0f * g[c];
          const float dist = (p == col) ? 0.f : sqrtf(fmaxf(d2, 0.f));
          g[c] = dist;
          s += dist;  // (the self pairs hold exact zeros; summation order = rank_blockmean's: a outer, c inner)
        }
      }
      const float mean = s / (float)(seq * seq - (I == J ? seq : 0));
      Dbar[row * b + J] = mean;
      mrow[J] = mean;
    }
    for (int c = B + lane; c < Bp; c += 64)  // padding columns of the own rows
      for (int a = 0; a < seq; ++a) G[(qb + I * seq + a) * Bp + c] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float dii = mrow[I];
    float loss = 0.f, cnt = 0.f;
    for (int J = lane; J < b; J += 64) {
      if (J == I) continue;
      const float h = dii - mrow[J] + margin;
      const bool on = h > 0.f;
      loss += on ? h : 0.f;
      cnt += on ? 1.f : 0.f;
      dDbar[row * b + J] = on ? -inv : 0.f;
    }
    loss = wave_sum(loss);
    cnt = wave_sum(cnt);
    if (lane == 0) {
      dDbar[row * b + I] = cnt * inv;
      diag_own[idx] = dii;
      diag_full[row] = dii;
      acc += loss * inv;
    }
    __builtin_amdgcn_wave_barrier();  // the LDS row is rewritten by the wave's next item
  }
  block_term_add(acc, terms + 3);
}

// ------------------------------------------------------------------------------------------------ launches
// The head is ~20 small dependent kernels on the step's serial path (both encoder streams wait for it): what it costs is launches,
// not bytes.  Kernels that do not depend on each other run as ONE launch -- a workgroup picks its part from blockIdx ranges --
// which leaves memset, pack, 3 products, rows | coefficients, 3 products, gradients: 11 launches instead of 22 (one rank).
struct NceGroup { int p0, nprob, width, gb0; float* Zn; float* dZn; float* nrm; float* S; float* lse; int blocks; };
struct HeadArgs {
  PairTable nce, orth;
  RankTable rk;
  NceGroup g[2];
  Shard sh;
  int seq, b, n2p, dim, B, Bp, Q, O, half;
  float margin, w_orth, w_rank, w_shared, w_private;
  float* X; float* sq; float* D; float* Dbar; float* dDbar; float* diag; float* rowsum; float* EX;
  float* chunk; float* pterms;   // this rank's exchange chunk / its partial-term slots
  const float* xall; float* terms;
  int blocks_rank, blocks_orth, blocks_total;  // parts three, four, five of the launch
  int fused_rank_rows;
};

__global__ __launch_bounds__(256) void head_pack_kernel(const HeadArgs a) {
  int bid = blockIdx.x;
  if (bid == 0 && threadIdx.x < 5) a.pterms[threadIdx.x] = 0.f;  // this rank's partial loss terms (nothing in this launch adds to them)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (bid < a.g[k].blocks) {
      nce_pack_body(bid, a.g[k].blocks, a.nce, a.g[k].p0, a.g[k].nprob, a.seq, a.b, a.n2p, a.dim, a.g[k].width, a.g[k].Zn, a.g[k].nrm, nullptr);
      return;
    }
    bid -= a.g[k].blocks;
  }
  rank_pack_body(bid, a.blocks_rank, a.rk, a.Q, a.B, a.Bp, a.dim, a.X, a.sq);
}

__global__ __launch_bounds__(256) void head_rows_kernel(const HeadArgs a) {
  int bid = blockIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (bid < a.g[k].blocks) {
      nce_rows_body(bid, a.g[k].blocks, a.nce, a.g[k].p0, a.g[k].nprob, a.seq, a.b, a.n2p, a.sh, a.g[k].gb0, a.g[k].S, a.chunk, a.g[k].lse, a.pterms);
      return;
    }
    bid -= a.g[k].blocks;
  }
  if (bid < a.blocks_rank) {
    rank_rows_body(bid, a.blocks_rank, a.Q, a.b, a.seq, a.B, a.Bp, a.margin, a.sh, a.D, a.sq, a.Dbar, a.dDbar, a.chunk + a.sh.o_diag, a.diag, a.pterms);
    return;
  }
  bid -= a.blocks_rank;
  orth_body(bid, a.blocks_orth, a.orth, a.O, a.B, a.dim, a.half, a.w_orth, a.sh.r0 * a.seq, a.sh.bl * a.seq, a.pterms);
}

// rows pass when the ranking rows are too long for rank_rows_body: InfoNCE rows + distances + orthogonality (block means and hinge follow)
__global__ __launch_bounds__(256) void head_rows_split_kernel(const HeadArgs a) {
  int bid = blockIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (bid < a.g[k].blocks) {
      nce_rows_body(bid, a.g[k].blocks, a.nce, a.g[k].p0, a.g[k].nprob, a.seq, a.b, a.n2p, a.sh, a.g[k].gb0, a.g[k].S, a.chunk, a.g[k].lse, a.pterms);
      return;
    }
    bid -= a.g[k].blocks;
  }
  if (bid < a.blocks_rank) {
    rank_dist_body(bid, a.blocks_rank, a.Q, a.B, a.Bp, a.sh.r0 * a.seq, a.sh.bl * a.seq, a.D, a.sq);
    return;
  }
  bid -= a.blocks_rank;
  orth_body(bid, a.blocks_orth, a.orth, a.O, a.B, a.dim, a.half, a.w_orth, a.sh.r0 * a.seq, a.sh.bl * a.seq, a.pterms);
}
__global__ __launch_bounds__(256) void rank_blockmean_kernel(int nq, int b, int seq, int Bp, Shard sh, const float* __restrict__ D, float* __restrict__ Dbar) {
  rank_blockmean_body(blockIdx.x, gridDim.x, nq, b, seq, Bp, sh, D, Dbar);
}
__global__ __launch_bounds__(256) void rank_hinge_kernel(int nq, int b, float margin, Shard sh, const float* __restrict__ Dbar, float* __restrict__ dDbar,
                                                         float* __restrict__ diag_own, float* __restrict__ diag_full, float* __restrict__ terms) {
  rank_hinge_body(blockIdx.x, gridDim.x, nq, b, margin, sh, Dbar, dDbar, diag_own, diag_full, terms);
}

__device__ __forceinline__ void loss_total_body(float* terms, const float* __restrict__ xall, const Shard& sh, float ws, float wp, float wo, float wr) {
  float t[4];
  for (int k = 0; k < 4; ++k) {
    t[k] = 0.f;
    for (int r = 0; r < sh.world; ++r) t[k] += xall[(long)r * sh.ch + sh.o_terms + k];
    terms[k] = t[k];
  }
  terms[4] = ws * t[0] + wp * t[1] + wo * t[2] + wr * t[3];
}

// coefficient pass: softmax weights of both InfoNCE groups, ranking coefficients, and (one thread of an extra workgroup) the loss terms
__global__ __launch_bounds__(256) void head_coeff_kernel(const HeadArgs a) {
  int bid = blockIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (bid < a.g[k].blocks) {
      nce_weights_body(bid, a.g[k].blocks, a.nce, a.g[k].p0, a.g[k].nprob, a.seq, a.b, a.n2p, a.sh, a.g[k].S, a.g[k].lse, a.w_shared, a.w_private);
      return;
    }
    bid -= a.g[k].blocks;
  }
  if (bid < a.blocks_rank) {
    rank_coeff_body(bid, a.blocks_rank, a.Q, a.b, a.seq, a.Bp, a.w_rank, a.margin, a.sh, a.D, a.Dbar, a.dDbar, a.diag, a.rowsum);
    return;
  }
  if (threadIdx.x == 0) loss_total_body(a.terms, a.xall, a.sh, a.w_shared, a.w_private, a.w_orth, a.w_rank);
}

// gradient pass: dL/dz of both InfoNCE groups (normalisation backward, scatter) and the ranking gradient
__global__ __launch_bounds__(256) void head_grad_kernel(const HeadArgs a) {
  int bid = blockIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (bid < a.g[k].blocks) {
      nce_unpack_body(bid, a.g[k].blocks, a.nce, a.g[k].p0, a.g[k].nprob, a.seq, a.b, a.n2p, a.dim, a.g[k].width, a.sh, a.g[k].Zn, a.g[k].nrm, a.g[k].dZn);
      return;
    }
    bid -= a.g[k].blocks;
  }
  rank_grad_body(bid, a.blocks_rank, a.rk, a.Q, a.Bp, a.dim, a.sh.r0 * a.seq, a.sh.bl * a.seq, a.X, a.rowsum, a.EX);
}

// ------------------------------------------------------------------------------------------------ host side
struct LossPlan {
  int b, n2, n2p, Bp, P_sh, P_pr, w_sh, w_pr, Q, O;
  size_t off_zn, off_nrm, off_S, off_lse, off_diag, off_dzn, off_X, off_sq, off_D, off_dbar, off_ddbar, off_rs, off_ex, off_xchg, total;
  Shard sh;
  int rank;
};

static int loss_plan(const focal_loss_desc* d, int rank, int world, LossPlan* pl) {
  FOCAL_CHECK_ARG(d != nullptr, "loss_head: null descriptor");
  FOCAL_CHECK_ARG(d->n_mod >= 1 && d->n_mod <= 4, "loss_head: n_mod=%d out of [1, 4]", d->n_mod);
  FOCAL_CHECK_ARG(d->seq >= 1 && d->B % d->seq == 0 && d->B / d->seq >= 2, "loss_head: batch %d must be >= 2 whole subsequences of %d", d->B, d->seq);
  FOCAL_CHECK_ARG(d->dim % 8 == 0 && d->dim <= 1024, "loss_head: dim %d unsupported", d->dim);
  const int M = d->n_mod;
  pl->b = d->B / d->seq;
  FOCAL_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "loss_head: rank %d of %d", rank, world);
  FOCAL_CHECK_ARG(pl->b % world == 0, "loss_head: %d subsequences do not split over %d ranks (every rank contributes the same number)", pl->b, world);
  pl->n2 = 2 * pl->b;
  pl->n2p = (pl->n2 + 3) & ~3;   // pitches of the similarity / distance matrices (16-byte chunks along the GEMM reduction index)
  pl->Bp = (d->B + 3) & ~3;
  pl->P_sh = 2 * (M * (M - 1) / 2);
  pl->P_pr = M;
  pl->w_sh = d->no_private ? d->dim : d->dim / 2;
  pl->w_pr = d->dim / 2;
  pl->Q = 2 * M;
  pl->O = 2 * (M + M * (M - 1) / 2);
  FOCAL_CHECK_ARG(pl->P_sh + pl->P_pr <= LOSS_MAXP && pl->O <= LOSS_MAXO && pl->Q <= LOSS_MAXQ, "loss_head: too many modality pairs");
  const size_t rows_sh = (size_t)pl->P_sh * d->seq * pl->n2p, rows_pr = (size_t)pl->P_pr * d->seq * pl->n2p;
  const size_t zn = rows_sh * pl->w_sh + rows_pr * pl->w_pr;
  // exchange chunk of one rank: lse of its rows of every (problem, step) block, its diagonal block means, its partial loss terms
  const int bl = pl->b / world;
  const long nblk = (long)(pl->P_sh + pl->P_pr) * d->seq;
  pl->rank = rank;
  pl->sh.r0 = rank * bl; pl->sh.bl = bl; pl->sh.world = world;
  pl->sh.o_diag = (int)(nblk * 2 * bl);
  pl->sh.o_terms = pl->sh.o_diag + pl->Q * bl;
  pl->sh.ch = (pl->sh.o_terms + 5 + 63) & ~63;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
  pl->off_zn = take(zn);
  pl->off_dzn = take(zn);
  pl->off_nrm = take(rows_sh + rows_pr);
  pl->off_lse = take(rows_sh + rows_pr);
  pl->off_diag = take((size_t)pl->Q * pl->b);
  pl->off_S = take((rows_sh + rows_pr) * pl->n2p);
  pl->off_X = take((size_t)pl->Q * pl->Bp * d->dim);
  pl->off_ex = take((size_t)pl->Q * pl->Bp * d->dim);
  pl->off_sq = take((size_t)pl->Q * pl->Bp);
  pl->off_rs = take((size_t)pl->Q * pl->Bp);
  pl->off_D = take((size_t)pl->Q * pl->Bp * pl->Bp);
  pl->off_dbar = take((size_t)pl->Q * pl->b * pl->b);
  pl->off_ddbar = take((size_t)pl->Q * pl->b * pl->b);
  pl->off_xchg = take((size_t)pl->sh.ch);  // the one-rank call's own chunk
  pl->total = o * sizeof(float);
  return FOCAL_OK;
}

extern "C" size_t focal_loss_head_workspace(const focal_loss_desc* d) {
  LossPlan pl;
  if (loss_plan(d, 0, 1, &pl) != FOCAL_OK) return 0;
  return pl.total;
}

extern "C" size_t focal_loss_head_exchange_floats(const focal_loss_desc* d, int world) {
  LossPlan pl;
  if (loss_plan(d, 0, world, &pl) != FOCAL_OK) return 0;
  return (size_t)pl.sh.ch;
}

static int f32_gemm(bool trb, int M, int N, int K, const float* A, long lda, long sA, const float* B, long ldb, long sB, float* C,
                    long ldc, long sC, int batch, float alpha, hipStream_t st) {
  GemmSpec s;
  s.compute = FOCAL_F32; s.a_dtype = FOCAL_F32; s.b_dtype = FOCAL_F32; s.c_dtype = FOCAL_F32;
  s.tra = false; s.trb = trb; s.proA = PRO_NONE; s.proB = PRO_NONE; s.epi = EPI_STORE;
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = M; p.N = N; p.K = K;
  p.A = A; p.lda = lda; p.strideA = sA;
  p.B = B; p.ldb = ldb; p.strideB = sB;
  p.C = C; p.ldc = ldc; p.strideC = sC;
  p.batch = batch; p.splits = 1; p.alpha = alpha;
  return focal_launch_gemm(s, p, st);
}

// The head's exact-fp32 products of one phase (similarity rows of both InfoNCE groups + Gram rows; coefficient rows x embeddings) do not
// depend on each other: ONE launch of the 64 x 64 tiles of focal_gemm_kernel behind a problem table instead of three (five when a
// rank's own rows are two runs per block).
#define LOSS_GEMM_MAX 6
struct GemmGroup { int n; int wg_end[LOSS_GEMM_MAX]; GemmParams p[LOSS_GEMM_MAX]; };
template <bool TRB> __global__ __launch_bounds__(256) void loss_gemm_group_kernel(const GemmGroup g) {
  const int b = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int q = 0; q < LOSS_GEMM_MAX - 1; ++q) pi += (q < g.n - 1 && b >= g.wg_end[q]) ? 1 : 0;
  const int start = pi > 0 ? g.wg_end[pi - 1] : 0;
  focal_gemm_body<float, float, float, float, false, TRB, PRO_NONE, PRO_NONE, EPI_STORE, 64, 64, 1>(g.p[pi], b - start, g.wg_end[pi] - start);
}
static void gemm_group_add(GemmGroup* g, int M, int N, int K, const float* A, long lda, long sA, const float* B, long ldb, long sB, float* C, long ldc,
                           long sC, int batch, float alpha) {
  GemmParams& p = g->p[g->n];
  memset(&p, 0, sizeof(p));
  p.M = M; p.N = N; p.K = K;
  p.A = A; p.lda = lda; p.strideA = sA;
  p.B = B; p.ldb = ldb; p.strideB = sB;
  p.C = C; p.ldc = ldc; p.strideC = sC;
  p.batch = batch; p.splits = 1; p.alpha = alpha;
  const int wgs = ceil_div(M, 64) * ceil_div(N, 64) * batch;
  g->wg_end[g->n] = (g->n ? g->wg_end[g->n - 1] : 0) + wgs;
  ++g->n;
}
static int gemm_group_launch(const GemmGroup& g, bool trb, hipStream_t st) {
  if (g.n == 0) return FOCAL_OK;
  const int wgs = g.wg_end[g.n - 1];
  if (trb) FOCAL_LAUNCH(loss_gemm_group_kernel<true>, dim3(wgs), dim3(256), 0, st, g);
  else FOCAL_LAUNCH(loss_gemm_group_kernel<false>, dim3(wgs), dim3(256), 0, st, g);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

struct LossTables { PairTable nce, orth; RankTable rk; };

static void loss_tables(const focal_loss_desc* d, const float* const* feats, float* const* dfeats, LossTables* t) {
  const int M = d->n_mod, half = d->dim / 2;
  memset(t, 0, sizeof(*t));
  // ---- problem tables (view-major feature order: index v*M + m)
  int np = 0;
  for (int v = 0; v < 2; ++v)  // shared family, loss.py:162-178
    for (int m1 = 0; m1 < M; ++m1)
      for (int m2 = m1 + 1; m2 < M; ++m2)
        t->nce.p[np++] = PairProb{feats[v * M + m1], feats[v * M + m2], dfeats[v * M + m1], dfeats[v * M + m2], 0, 0, 0};
  for (int m = 0; m < M; ++m)  // private family, loss.py:181-186
    t->nce.p[np++] = PairProb{feats[m], feats[M + m], dfeats[m], dfeats[M + m], half, half, 1};
  int no = 0;
  for (int v = 0; v < 2; ++v)  // loss.py:195-209
    for (int m = 0; m < M; ++m) {
      t->orth.p[no++] = PairProb{feats[v * M + m], feats[v * M + m], dfeats[v * M + m], dfeats[v * M + m], 0, half, 2};
      for (int m2 = m + 1; m2 < M; ++m2)
        t->orth.p[no++] = PairProb{feats[v * M + m], feats[v * M + m2], dfeats[v * M + m], dfeats[v * M + m2], half, half, 2};
    }
  for (int i = 0; i < 2 * M; ++i) { t->rk.x[i] = feats[i]; t->rk.d[i] = dfeats[i]; }
}

// The rows a rank owns in an [n2p]-row block are the two runs h * b + r0 .. + bl - 1 (h = view half): a product over own rows is one
// batched GEMM per half (all ranks of one: the two runs are adjacent and cover the block, one GEMM as before).
template <typename F> static int over_own_halves(const LossPlan& pl, F&& f) {
  if (pl.sh.world == 1) return f(0, pl.n2p);
  for (int h = 0; h < 2; ++h)
    if (int rc = f(h * pl.b + pl.sh.r0, pl.sh.bl)) return rc;
  return FOCAL_OK;
}

// the argument block of the fused launches (block counts are filled in per launch)
static void head_args(const focal_loss_desc* d, const LossPlan& pl, const float* const* feats, float* const* dfeats, float* chunk,
                      const float* xall, float* terms, float* ws, HeadArgs* a) {
  LossTables tb;
  loss_tables(d, feats, dfeats, &tb);
  memset(a, 0, sizeof(*a));
  a->nce = tb.nce; a->orth = tb.orth; a->rk = tb.rk;
  a->sh = pl.sh;
  a->seq = d->seq; a->b = pl.b; a->n2p = pl.n2p; a->dim = d->dim; a->B = d->B; a->Bp = pl.Bp; a->Q = pl.Q; a->O = pl.O; a->half = d->dim / 2;
  a->margin = d->margin; a->w_orth = d->w_orth; a->w_rank = d->w_rank; a->w_shared = d->w_shared; a->w_private = d->w_private;
  a->X = ws + pl.off_X; a->sq = ws + pl.off_sq; a->D = ws + pl.off_D; a->Dbar = ws + pl.off_dbar; a->dDbar = ws + pl.off_ddbar;
  a->diag = ws + pl.off_diag; a->rowsum = ws + pl.off_rs; a->EX = ws + pl.off_ex;
  a->chunk = chunk; a->pterms = chunk ? chunk + pl.sh.o_terms : nullptr;
  a->xall = xall; a->terms = terms;
  // the two InfoNCE families may have different widths (tag == "noPrivate"), so they are two groups of problems
  size_t zoff = 0, roff = 0;
  int gb0 = 0;
  for (int grp = 0; grp < 2; ++grp) {
    NceGroup& g = a->g[grp];
    g.p0 = grp == 0 ? 0 : pl.P_sh; g.nprob = grp == 0 ? pl.P_sh : pl.P_pr; g.width = grp == 0 ? pl.w_sh : pl.w_pr;
    const long rows = (long)g.nprob * d->seq * pl.n2p;
    g.gb0 = gb0;
    g.Zn = ws + pl.off_zn + zoff; g.dZn = ws + pl.off_dzn + zoff; g.nrm = ws + pl.off_nrm + roff;
    g.S = ws + pl.off_S + roff * pl.n2p; g.lse = ws + pl.off_lse + roff;
    zoff += (size_t)rows * g.width;
    roff += rows;
    gb0 += g.nprob * d->seq;
  }
  static const bool split_rows = getenv("FOCAL_LOSS_RANK_SPLIT") != nullptr;  // (tests: the three-pass form of longer rows on small inputs)
  a->fused_rank_rows = pl.b <= RANK_ROW_MAX_B && !split_rows;
}
static inline int capped(long blocks, int cap) { return (int)(blocks > cap ? cap : (blocks < 1 ? 1 : blocks)); }

// Phase A: everything up to the quantities other ranks need -- zero the gradients, InfoNCE similarity rows + lse, the ranking
// distances, block means and hinge rows, the orthogonality term; this rank's lse / diagonal means / partial terms -> `chunk`.
static int loss_phase_a(const focal_loss_desc* d, const LossPlan& pl, const float* const* feats, float* terms, float* const* dfeats,
                        float* chunk, float* ws, hipStream_t st) {
  const int M = d->n_mod, B = d->B, dim = d->dim, seq = d->seq, n2p = pl.n2p, Bp = pl.Bp;
  const Shard sh = pl.sh;
  // the binding hands over ONE allocation [2M gradients | terms]: a single memset node instead of 2M + 1
  const size_t gbytes = (size_t)B * dim * sizeof(float);
  // (any order of the 2M blocks inside it: the binding places a modality's two views next to each other)
  float* lo = dfeats[0];
  for (int i = 1; i < 2 * M; ++i) lo = dfeats[i] < lo ? dfeats[i] : lo;
  bool packed = true;
  unsigned seen = 0;
  for (int i = 0; i < 2 * M; ++i) {
    const size_t off = (size_t)(dfeats[i] - lo), slot = off / ((size_t)B * dim);
    packed = packed && (off % ((size_t)B * dim) == 0) && slot < (size_t)(2 * M) && !(seen & (1u << slot));
    if (slot < 32) seen |= 1u << slot;
  }
  packed = packed && (terms == lo + (size_t)2 * M * B * dim);
  if (packed) {
    (void)hipMemsetAsync(lo, 0, 2 * M * gbytes + 5 * sizeof(float), st);
  } else {
    (void)hipMemsetAsync(terms, 0, 5 * sizeof(float), st);
    for (int i = 0; i < 2 * M; ++i) (void)hipMemsetAsync(dfeats[i], 0, gbytes, st);
  }
  HeadArgs a;
  head_args(d, pl, feats, dfeats, chunk, nullptr, terms, ws, &a);
  const int Q = pl.Q, rs0 = sh.r0 * seq, nr = sh.bl * seq;

  // ---- launch 1: normalised InfoNCE rows of both groups, ranking rows + squared norms (this rank's partial terms zeroed)
  for (int k = 0; k < 2; ++k) a.g[k].blocks = a.g[k].nprob ? ceil_div((long)a.g[k].nprob * seq * n2p, 4) : 0;
  a.blocks_rank = ceil_div((long)Q * Bp, 4);
  FOCAL_LAUNCH(head_pack_kernel, dim3(a.g[0].blocks + a.g[1].blocks + a.blocks_rank), dim3(256), 0, st, a);

  // ---- products: similarity rows S = Zn Zn^T / T per group, Gram rows G = X X^T (exact-fp32 MFMA)
  static const bool one_gemm_launch = getenv("FOCAL_LOSS_GEMM_SEPARATE") == nullptr;
  GemmGroup gg;
  gg.n = 0;
  for (int k = 0; k < 2; ++k) {
    const NceGroup& g = a.g[k];
    if (g.nprob == 0) continue;
    if (int rc = over_own_halves(pl, [&](int row0, int nrow) {
          if (one_gemm_launch) {
            gemm_group_add(&gg, nrow, n2p, g.width, g.Zn + (long)row0 * g.width, g.width, (long)n2p * g.width, g.Zn, g.width, (long)n2p * g.width,
                           g.S + (long)row0 * n2p, n2p, (long)n2p * n2p, g.nprob * seq, 1.0f / d->temperature);
            return (int)FOCAL_OK;
          }
          return f32_gemm(false, nrow, n2p, g.width, g.Zn + (long)row0 * g.width, g.width, (long)n2p * g.width, g.Zn, g.width, (long)n2p * g.width,
                          g.S + (long)row0 * n2p, n2p, (long)n2p * n2p, g.nprob * seq, 1.0f / d->temperature, st);
        })) return rc;
  }
  const int mrows = sh.world == 1 ? Bp : nr;
  if (one_gemm_launch) {
    gemm_group_add(&gg, mrows, Bp, dim, a.X + (long)rs0 * dim, dim, (long)Bp * dim, a.X, dim, (long)Bp * dim, a.D + (long)rs0 * Bp, Bp, (long)Bp * Bp, Q, 1.0f);
    if (int rc = gemm_group_launch(gg, false, st)) return rc;
  } else if (int rc = f32_gemm(false, mrows, Bp, dim, a.X + (long)rs0 * dim, dim, (long)Bp * dim, a.X, dim, (long)Bp * dim, a.D + (long)rs0 * Bp, Bp,
                               (long)Bp * Bp, Q, 1.0f, st)) return rc;

  // ---- launch 2: lse rows + InfoNCE terms, ranking distances / block means / hinges (loss.py:189-192), orthogonality (row-local)
  for (int k = 0; k < 2; ++k) a.g[k].blocks = a.g[k].nprob ? loss_row_blocks((long)a.g[k].nprob * seq * 2 * sh.bl) : 0;
  a.blocks_orth = loss_row_blocks((long)pl.O * nr);
  if (a.fused_rank_rows) {
    a.blocks_rank = loss_row_blocks((long)Q * sh.bl);
    FOCAL_LAUNCH(head_rows_kernel, dim3(a.g[0].blocks + a.g[1].blocks + a.blocks_rank + a.blocks_orth), dim3(256), 0, st, a);
  } else {
    a.blocks_rank = capped(ceil_div((long)Q * nr * Bp, 256), 8192);
    FOCAL_LAUNCH(head_rows_split_kernel, dim3(a.g[0].blocks + a.g[1].blocks + a.blocks_rank + a.blocks_orth), dim3(256), 0, st, a);
    FOCAL_LAUNCH(rank_blockmean_kernel, dim3(capped(ceil_div((long)Q * sh.bl * pl.b, 256), 4096)), dim3(256), 0, st, Q, pl.b, seq, Bp, sh, a.D, a.Dbar);
    FOCAL_LAUNCH(rank_hinge_kernel, dim3(loss_row_blocks((long)Q * sh.bl)), dim3(256), 0, st, Q, pl.b, d->margin, sh, a.Dbar, a.dDbar,
                       chunk + sh.o_diag, a.diag, a.pterms);
  }
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

// Phase B: `xall` = the chunks of all ranks in rank order.  Softmax coefficient rows and dL/dz of the own InfoNCE rows, ranking
// coefficients and gradient of the own samples, the loss terms.
static int loss_phase_b(const focal_loss_desc* d, const LossPlan& pl, const float* const* feats, float* terms, float* const* dfeats,
                        const float* xall, float* ws, hipStream_t st) {
  const int dim = d->dim, seq = d->seq, b = pl.b, n2 = pl.n2, n2p = pl.n2p, Bp = pl.Bp;
  const Shard sh = pl.sh;
  HeadArgs a;
  head_args(d, pl, feats, dfeats, nullptr, xall, terms, ws, &a);
  if (sh.world > 1) {  // (one rank: phase A has written every row's lse / diagonal mean in place already)
    const int nblk = (pl.P_sh + pl.P_pr) * seq;
    FOCAL_LAUNCH(xchg_unpack_kernel, dim3(capped(ceil_div((long)nblk * n2 + (long)pl.Q * b, 256), 1024)), dim3(256), 0, st, sh, nblk, b, n2p, pl.Q,
                       xall, ws + pl.off_lse, ws + pl.off_diag);
  }
  const int Q = pl.Q, rs0 = sh.r0 * seq, nr = sh.bl * seq;
  // ---- launch 3: softmax coefficients of both groups (in place of S), ranking coefficients (in place of D) + their row sums, loss terms
  for (int k = 0; k < 2; ++k) a.g[k].blocks = a.g[k].nprob ? capped(ceil_div((long)a.g[k].nprob * seq * 2 * sh.bl * n2p, 256), 8192) : 0;
  a.blocks_rank = ceil_div((long)Q * nr, 4);
  FOCAL_LAUNCH(head_coeff_kernel, dim3(a.g[0].blocks + a.g[1].blocks + a.blocks_rank + 1), dim3(256), 0, st, a);
  // ---- products: dZn = W Zn per group, EX = E X
  static const bool one_gemm_launch = getenv("FOCAL_LOSS_GEMM_SEPARATE") == nullptr;
  GemmGroup gg;
  gg.n = 0;
  for (int k = 0; k < 2; ++k) {
    const NceGroup& g = a.g[k];
    if (g.nprob == 0) continue;
    if (int rc = over_own_halves(pl, [&](int row0, int nrow) {
          if (one_gemm_launch) {
            gemm_group_add(&gg, nrow, g.width, n2p, g.S + (long)row0 * n2p, n2p, (long)n2p * n2p, g.Zn, g.width, (long)n2p * g.width,
                           g.dZn + (long)row0 * g.width, g.width, (long)n2p * g.width, g.nprob * seq, 1.0f / ((float)seq * n2 * d->temperature));
            return (int)FOCAL_OK;
          }
          return f32_gemm(true, nrow, g.width, n2p, g.S + (long)row0 * n2p, n2p, (long)n2p * n2p, g.Zn, g.width, (long)n2p * g.width,
                          g.dZn + (long)row0 * g.width, g.width, (long)n2p * g.width, g.nprob * seq, 1.0f / ((float)seq * n2 * d->temperature), st);
        })) return rc;
  }
  if (one_gemm_launch) {
    gemm_group_add(&gg, nr, dim, Bp, a.D + (long)rs0 * Bp, Bp, (long)Bp * Bp, a.X, dim, (long)Bp * dim, a.EX + (long)rs0 * dim, dim, (long)Bp * dim, Q, 1.0f);
    if (int rc = gemm_group_launch(gg, true, st)) return rc;
  } else if (int rc = f32_gemm(true, nr, dim, Bp, a.D + (long)rs0 * Bp, Bp, (long)Bp * Bp, a.X, dim, (long)Bp * dim, a.EX + (long)rs0 * dim, dim,
                               (long)Bp * dim, Q, 1.0f, st)) return rc;
  // ---- launch 4: gradients back onto the embeddings
  for (int k = 0; k < 2; ++k) a.g[k].blocks = a.g[k].nprob ? ceil_div((long)a.g[k].nprob * seq * 2 * sh.bl, 4) : 0;
  a.blocks_rank = capped(ceil_div((long)Q * nr * dim, 256), 8192);
  FOCAL_LAUNCH(head_grad_kernel, dim3(a.g[0].blocks + a.g[1].blocks + a.blocks_rank), dim3(256), 0, st, a);
  FOCAL_LAUNCH_CHECK();
  return FOCAL_OK;
}

static int loss_args(const focal_loss_desc* d, int rank, int world, const float* const* feats, float* terms, float* const* dfeats,
                     void* workspace, size_t workspace_bytes, LossPlan* pl) {
  if (int rc = loss_plan(d, rank, world, pl)) return rc;
  FOCAL_CHECK_ARG(feats && terms && dfeats && workspace, "loss_head: null argument");
  if (workspace_bytes < pl->total) {
    focal_set_error("loss_head: workspace %zu < required %zu bytes", workspace_bytes, pl->total);
    return FOCAL_EWORKSPACE;
  }
  return FOCAL_OK;
}

extern "C" int focal_loss_head(const focal_loss_desc* d, const float* const* feats, float* terms, float* const* dfeats,
                               void* workspace, size_t workspace_bytes, void* stream) {
  LossPlan pl;
  if (int rc = loss_args(d, 0, 1, feats, terms, dfeats, workspace, workspace_bytes, &pl)) return rc;
  float* ws = reinterpret_cast<float*>(workspace);
  float* chunk = ws + pl.off_xchg;
  if (int rc = loss_phase_a(d, pl, feats, terms, dfeats, chunk, ws, (hipStream_t)stream)) return rc;
  return loss_phase_b(d, pl, feats, terms, dfeats, chunk, ws, (hipStream_t)stream);
}

extern "C" int focal_loss_head_shard_a(const focal_loss_desc* d, int rank, int world, const float* const* feats, float* terms,
                                       float* const* dfeats, float* chunk, void* workspace, size_t workspace_bytes, void* stream) {
  LossPlan pl;
  if (int rc = loss_args(d, rank, world, feats, terms, dfeats, workspace, workspace_bytes, &pl)) return rc;
  FOCAL_CHECK_ARG(chunk != nullptr, "loss_head_shard_a: null exchange chunk");
  return loss_phase_a(d, pl, feats, terms, dfeats, chunk, reinterpret_cast<float*>(workspace), (hipStream_t)stream);
}

extern "C" int focal_loss_head_shard_b(const focal_loss_desc* d, int rank, int world, const float* const* feats, float* terms,
                                       float* const* dfeats, const float* chunks, void* workspace, size_t workspace_bytes, void* stream) {
  LossPlan pl;
  if (int rc = loss_args(d, rank, world, feats, terms, dfeats, workspace, workspace_bytes, &pl)) return rc;
  FOCAL_CHECK_ARG(chunks != nullptr, "loss_head_shard_b: null exchange buffer");
  return loss_phase_b(d, pl, feats, terms, dfeats, chunks, reinterpret_cast<float*>(workspace), (hipStream_t)stream);
}
