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
#include "gemm.hpp"

#define LOSS_MAXP 32   // InfoNCE problems  (2 views x mod pairs + mods)
#define LOSS_MAXQ 8    // ranking problems  (2 views x mods)
#define LOSS_MAXO 32   // orthogonality problems

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
__global__ __launch_bounds__(256) void nce_pack_kernel(PairTable tab, int p0, int nprob, int seq, int b, int n2p, int dim, int width,
                                                       float* __restrict__ Zn, float* __restrict__ nrm, float* __restrict__ zero5) {
  const int lane = threadIdx.x & 63;
  if (zero5 != nullptr && blockIdx.x == 0 && threadIdx.x < 5) zero5[threadIdx.x] = 0.f;  // this rank's partial loss terms (first launch of the head)
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
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
__global__ __launch_bounds__(256) void nce_rows_kernel(PairTable tab, int p0, int nprob, int seq, int b, int n2p, Shard sh, int gb0,
                                                       const float* __restrict__ S, float* __restrict__ lse_own, float* __restrict__ lse_full,
                                                       float* __restrict__ terms) {
  const int lane = threadIdx.x & 63;
  const int n2 = 2 * b, own = 2 * sh.bl;
  const long rows = (long)nprob * seq * own;
  float acc0 = 0.f, acc1 = 0.f;  // contributions to terms[0] (shared family) / terms[1] (private family)
  for (long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6); idx < rows; idx += (long)gridDim.x * 4) {
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
__global__ __launch_bounds__(256) void nce_weights_kernel(PairTable tab, int p0, int nprob, int seq, int b, int n2p, Shard sh,
                                                          float* __restrict__ S, const float* __restrict__ lse, float w_shared, float w_private) {
  const int n2 = 2 * b, own = 2 * sh.bl;
  const long total = (long)nprob * seq * own * n2p;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
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
__global__ __launch_bounds__(256) void nce_unpack_kernel(PairTable tab, int p0, int nprob, int seq, int b, int n2p, int dim, int width, Shard sh,
                                                         const float* __restrict__ Zn, const float* __restrict__ nrm,
                                                         const float* __restrict__ dZn) {
  const int lane = threadIdx.x & 63, own = 2 * sh.bl;
  const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
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
__global__ __launch_bounds__(256) void rank_pack_kernel(RankTable tab, int nq, int B, int Bp, int dim, float* __restrict__ X,
                                                        float* __restrict__ sq) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
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
__global__ __launch_bounds__(256) void rank_dist_kernel(int nq, int B, int Bp, int rs0, int nr, float* __restrict__ G, const float* __restrict__ sq) {
  const long total = (long)nq * nr * Bp;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = e % Bp;
    const int r = rs0 + (int)((e / Bp) % nr);
    const long qb = (e / ((long)Bp * nr)) * Bp;
    const long at = (qb + r) * Bp + c;
    const float d2 = sq[qb + r] + sq[qb + c] - 2.0f * G[at];
    G[at] = (r == c || c >= B) ? 0.f : sqrtf(fmaxf(d2, 0.f));
  }
}

// Dbar[q][I][J] = mean of the seq x seq block (self pairs excluded), loss.py:118-124; own subsequences I, all J
__global__ __launch_bounds__(256) void rank_blockmean_kernel(int nq, int b, int seq, int Bp, Shard sh, const float* __restrict__ D, float* __restrict__ Dbar) {
  const long total = (long)nq * sh.bl * b;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
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
__global__ __launch_bounds__(256) void rank_hinge_kernel(int nq, int b, float margin, Shard sh, const float* __restrict__ Dbar,
                                                         float* __restrict__ dDbar, float* __restrict__ diag_own, float* __restrict__ diag_full,
                                                         float* __restrict__ terms) {
  const int lane = threadIdx.x & 63;
  const float inv = 1.0f / ((float)b * (float)(b - 1));
  float acc = 0.f;
  for (long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6); idx < (long)nq * sh.bl; idx += (long)gridDim.x * 4) {
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
__global__ __launch_bounds__(256) void rank_coeff_kernel(int nq, int b, int seq, int Bp, float w_rank, float margin, Shard sh, float* __restrict__ D,
                                                         const float* __restrict__ Dbar, const float* __restrict__ dDbar,
                                                         const float* __restrict__ diag, float* __restrict__ rowsum) {
  const int lane = threadIdx.x & 63;
  const int B = b * seq, nr = sh.bl * seq;
  const float inv = 1.0f / ((float)b * (float)(b - 1));
  const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
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

__global__ __launch_bounds__(256) void rank_grad_kernel(RankTable tab, int nq, int Bp, int dim, int rs0, int nr, const float* __restrict__ X,
                                                        const float* __restrict__ rowsum, const float* __restrict__ EX) {
  const long total = (long)nq * nr * dim;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = e % dim, s = rs0 + (int)((e / dim) % nr), q = e / ((long)dim * nr);
    const long at = ((long)q * Bp + s) * dim + c;
    atomicAdd(tab.d[q] + (long)s * dim + c, rowsum[(long)q * Bp + s] * X[at] - EX[at]);
  }
}

// ------------------------------------------------------------------------------------------------ orthogonality
// CosineEmbeddingLoss(target = -1, margin 0, mean) = mean(max(0, cos)), loss.py:89-106; one wave per sample.
__global__ __launch_bounds__(256) void orth_kernel(PairTable tab, int nprob, int B, int dim, int width, float w_orth, int rs0, int nr,
                                                   float* __restrict__ terms) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6); idx < (long)nprob * nr; idx += (long)gridDim.x * 4) {
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

// terms[0..3] = sum over ranks of the partial terms in the gathered chunks, terms[4] = their weighted sum
__global__ void loss_total_kernel(float* terms, const float* __restrict__ xall, Shard sh, float ws, float wp, float wo, float wr) {
  float t[4];
  for (int k = 0; k < 4; ++k) {
    t[k] = 0.f;
    for (int r = 0; r < sh.world; ++r) t[k] += xall[(long)r * sh.ch + sh.o_terms + k];
    terms[k] = t[k];
  }
  terms[4] = ws * t[0] + wp * t[1] + wo * t[2] + wr * t[3];
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

// Phase A: everything up to the quantities other ranks need -- zero the gradients, InfoNCE similarity rows + lse, the ranking
// distances, block means and hinge rows, the orthogonality term; this rank's lse / diagonal means / partial terms -> `chunk`.
static int loss_phase_a(const focal_loss_desc* d, const LossPlan& pl, const float* const* feats, float* terms, float* const* dfeats,
                        float* chunk, float* ws, hipStream_t st) {
  const int M = d->n_mod, B = d->B, dim = d->dim, seq = d->seq, b = pl.b, n2p = pl.n2p, Bp = pl.Bp, half = d->dim / 2;
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
  float* pterms = chunk + sh.o_terms;  // partial terms of this rank (zeroed by the first pack launch)
  bool zeroed = false;
  LossTables tb;
  loss_tables(d, feats, dfeats, &tb);

  // ---- InfoNCE: the two families may have different widths (tag == "noPrivate"), so they run as two groups
  size_t zoff = 0, roff = 0;
  int gb0 = 0;
  for (int grp = 0; grp < 2; ++grp) {
    const int p0 = grp == 0 ? 0 : pl.P_sh, nprob = grp == 0 ? pl.P_sh : pl.P_pr, width = grp == 0 ? pl.w_sh : pl.w_pr;
    if (nprob == 0) continue;
    const long rows = (long)nprob * seq * n2p, own_rows = (long)nprob * seq * 2 * sh.bl;
    float* Zn = ws + pl.off_zn + zoff;
    float* nrm = ws + pl.off_nrm + roff;
    float* S = ws + pl.off_S + roff * n2p;
    hipLaunchKernelGGL(nce_pack_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, tb.nce, p0, nprob, seq, b, n2p, dim, width, Zn, nrm,
                       zeroed ? nullptr : pterms);
    zeroed = true;
    if (int rc = over_own_halves(pl, [&](int row0, int nrow) {
          return f32_gemm(false, nrow, n2p, width, Zn + (long)row0 * width, width, (long)n2p * width, Zn, width, (long)n2p * width,
                          S + (long)row0 * n2p, n2p, (long)n2p * n2p, nprob * seq, 1.0f / d->temperature, st);
        })) return rc;
    hipLaunchKernelGGL(nce_rows_kernel, dim3(loss_row_blocks(own_rows)), dim3(256), 0, st, tb.nce, p0, nprob, seq, b, n2p, sh, gb0, S, chunk,
                       ws + pl.off_lse + roff, pterms);
    zoff += (size_t)rows * width;
    roff += rows;
    gb0 += nprob * seq;
  }

  // ---- temporal ranking, loss.py:189-192: distances, block means and hinges of the own rows
  {
    const int Q = pl.Q, rs0 = sh.r0 * seq, nr = sh.bl * seq;
    float* X = ws + pl.off_X; float* sq = ws + pl.off_sq;
    float* D = ws + pl.off_D; float* Dbar = ws + pl.off_dbar; float* dDbar = ws + pl.off_ddbar;
    hipLaunchKernelGGL(rank_pack_kernel, dim3(ceil_div((long)Q * Bp, 4)), dim3(256), 0, st, tb.rk, Q, B, Bp, dim, X, sq);
    const int mrows = sh.world == 1 ? Bp : nr;
    if (int rc = f32_gemm(false, mrows, Bp, dim, X + (long)rs0 * dim, dim, (long)Bp * dim, X, dim, (long)Bp * dim, D + (long)rs0 * Bp, Bp,
                          (long)Bp * Bp, Q, 1.0f, st)) return rc;
    int eb = ceil_div((long)Q * nr * Bp, 256);
    if (eb > 8192) eb = 8192;
    hipLaunchKernelGGL(rank_dist_kernel, dim3(eb), dim3(256), 0, st, Q, B, Bp, rs0, nr, D, sq);
    int bb = ceil_div((long)Q * sh.bl * b, 256);
    if (bb > 4096) bb = 4096;
    hipLaunchKernelGGL(rank_blockmean_kernel, dim3(bb), dim3(256), 0, st, Q, b, seq, Bp, sh, D, Dbar);
    hipLaunchKernelGGL(rank_hinge_kernel, dim3(loss_row_blocks((long)Q * sh.bl)), dim3(256), 0, st, Q, b, d->margin, sh, Dbar, dDbar,
                       chunk + sh.o_diag, ws + pl.off_diag, pterms);
    // ---- orthogonality (row-local)
    hipLaunchKernelGGL(orth_kernel, dim3(loss_row_blocks((long)pl.O * nr)), dim3(256), 0, st, tb.orth, pl.O, B, dim, half, d->w_orth, rs0, nr, pterms);
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
  LossTables tb;
  loss_tables(d, feats, dfeats, &tb);
  if (sh.world > 1) {  // (one rank: phase A has written every row's lse / diagonal mean in place already)
    const int nblk = (pl.P_sh + pl.P_pr) * seq;
    int ub = ceil_div((long)nblk * n2 + (long)pl.Q * b, 256);
    if (ub > 1024) ub = 1024;
    hipLaunchKernelGGL(xchg_unpack_kernel, dim3(ub), dim3(256), 0, st, sh, nblk, b, n2p, pl.Q, xall, ws + pl.off_lse, ws + pl.off_diag);
  }
  size_t zoff = 0, roff = 0;
  for (int grp = 0; grp < 2; ++grp) {
    const int p0 = grp == 0 ? 0 : pl.P_sh, nprob = grp == 0 ? pl.P_sh : pl.P_pr, width = grp == 0 ? pl.w_sh : pl.w_pr;
    if (nprob == 0) continue;
    const long rows = (long)nprob * seq * n2p, own_rows = (long)nprob * seq * 2 * sh.bl;
    float* Zn = ws + pl.off_zn + zoff;
    float* dZn = ws + pl.off_dzn + zoff;
    float* nrm = ws + pl.off_nrm + roff;
    float* S = ws + pl.off_S + roff * n2p;
    int eb = ceil_div(own_rows * n2p, 256);
    if (eb > 8192) eb = 8192;
    hipLaunchKernelGGL(nce_weights_kernel, dim3(eb), dim3(256), 0, st, tb.nce, p0, nprob, seq, b, n2p, sh, S, ws + pl.off_lse + roff, d->w_shared,
                       d->w_private);
    if (int rc = over_own_halves(pl, [&](int row0, int nrow) {
          return f32_gemm(true, nrow, width, n2p, S + (long)row0 * n2p, n2p, (long)n2p * n2p, Zn, width, (long)n2p * width,
                          dZn + (long)row0 * width, width, (long)n2p * width, nprob * seq, 1.0f / ((float)seq * n2 * d->temperature), st);
        })) return rc;
    hipLaunchKernelGGL(nce_unpack_kernel, dim3(ceil_div(own_rows, 4)), dim3(256), 0, st, tb.nce, p0, nprob, seq, b, n2p, dim, width, sh, Zn, nrm, dZn);
    zoff += (size_t)rows * width;
    roff += rows;
  }
  {
    const int Q = pl.Q, rs0 = sh.r0 * seq, nr = sh.bl * seq;
    float* X = ws + pl.off_X; float* EX = ws + pl.off_ex; float* rs = ws + pl.off_rs;
    float* D = ws + pl.off_D; float* Dbar = ws + pl.off_dbar; float* dDbar = ws + pl.off_ddbar;
    hipLaunchKernelGGL(rank_coeff_kernel, dim3(ceil_div((long)Q * nr, 4)), dim3(256), 0, st, Q, b, seq, Bp, d->w_rank, d->margin, sh, D, Dbar, dDbar,
                       ws + pl.off_diag, rs);
    if (int rc = f32_gemm(true, nr, dim, Bp, D + (long)rs0 * Bp, Bp, (long)Bp * Bp, X, dim, (long)Bp * dim, EX + (long)rs0 * dim, dim,
                          (long)Bp * dim, Q, 1.0f, st)) return rc;
    int gb = ceil_div((long)Q * nr * dim, 256);
    if (gb > 8192) gb = 8192;
    hipLaunchKernelGGL(rank_grad_kernel, dim3(gb), dim3(256), 0, st, tb.rk, Q, Bp, dim, rs0, nr, X, rs, EX);
  }
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(1), 0, st, terms, xall, sh, d->w_shared, d->w_private, d->w_orth, d->w_rank);
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
