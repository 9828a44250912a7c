/* C ABI of libmpnn_hip.so: gfx950 (MI355X) kernels for the ProteinMPNN message-passing hot path.
 *
 * The reference (FridrichMethod/MPNN) has no native plugin ABI: its hot path is the Python
 * nn.Module `ProteinMPNN.forward` in src/mpnn/models/model_utils.py, dispatching to ATen ops.
 * Each entry point below replaces the ATen op sequence of the cited reference lines; the
 * Python module `mpnn_amd` (same constructor / forward / state_dict as the reference) binds
 * them with ctypes.  All citations are to src/mpnn/models/model_utils.py.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (hipMalloc or a
 *     torch tensor).  The library never allocates, frees or retains pointers and never synchronises.
 *   - `stream` is a hipStream_t passed as void*; kernels are launched on it only.
 *   - return 0 on success; -1 for rejected arguments; otherwise the hipError_t of the launch.
 *     mpnn_last_error() returns a thread-local message.  No C++ exception crosses the ABI.
 *   - fp32 everywhere; hidden size is fixed at 128 (the reference's only used value), K <= 64.
 *   - "packed" weights are in the MFMA A-fragment layouts produced by mpnn_pack_linear_split (f16 hi/lo halves: every
 *     128x128 matrix of the edge / node kernels) or mpnn_pack_linear (fp32: the two small heads).
 *   - "hE_frag" is the edge tensor in fragment layout [node][8][KT][64][4] floats, KT = ceil(K/16):
 *       hE_frag[n][t][c][lane][r] = h_E[n, 16c + (lane&15), 16t + 4(lane>>4) + r]
 *     (size n_nodes * 8 * KT * 256 floats).  Convert with mpnn_edges_to/from_natural.
 *   - E_idx32 is the neighbour list as int32 [node][16*KT], slots >= K filled with the node itself.
 */
#ifndef MPNN_HIP_H
#define MPNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int mpnn_version(void);

/* Recommended workspace size in bytes for the entry points that take one ("mpnn_edge_wgrad": n_rows = residues;
 * "mpnn_node_wgrad" / "mpnn_node_wgrad_batch": n_rows = rows of the operands, n_blocks = number of 128 x 128 blocks of
 * all outputs of the call; smaller workspaces are accepted down to one partial (16512 floats) and cost parallelism).
 * The backward / loss / norm entry points that reduce per-workgroup partial sums in a fixed order need theirs in full:
 * "mpnn_enc_edge_update_bwd", "mpnn_edge_features_bwd" (fixed), "mpnn_node_update_bwd", "mpnn_logits_log_softmax_bwd",
 * "mpnn_loss_smoothed" (n_rows = residues), "mpnn_gather_rows" (n_blocks = max_chunks), "mpnn_grad_sumsq" (n_rows =
 * elements of the largest tensor, n_blocks = tensors).
 * 0 for the entry points that need none, -1 for an unknown name.  Host-only: touches no device.
 * (SURVEY section 8b lists this query in the build-defined ABI; nothing in the reference corresponds to it.) */
long long mpnn_workspace_bytes(const char* entry, int n_rows, int n_blocks);
const char* mpnn_last_error(void);
/* 1 when the library was built with one of the timing-only ablation switches of csrc/mpnn_ablation.h (its results are
 * wrong by design; the Python binding refuses such a build unless MPNN_ALLOW_ABLATION=1), else 0.  No reference
 * counterpart: build hygiene. */
int mpnn_ablation_build(void);

/* nn.Linear weight W[out_f, ld] columns [col0, col0+in_f) -> packed[ceil(out/16)][ceil(in/16)][64][4],
 * zero padded.  Weight layout prep for every Linear on the path (:64-69, :112-114, :147-148, :212, :400, :412). */
int mpnn_pack_linear(const float* W, int out_f, int in_f, int ld, int col0, float* packed, void* stream);

/* Same, for the split-precision MFMA path (mpnn_common.h): every weight is stored as f16 hi + f16 lo
 * (22 significant bits), packed[ceil(out/16)][ceil(in/32)][2][64][8 halfs].  A product a*b is evaluated on the
 * f16 matrix pipe as a_lo b_hi + a_hi b_lo + a_hi b_hi with fp32 accumulation (relative error <= ~2^-21).
 * rot rotates the input features (packed column c = source column (c + rot) mod in_f): the featurizer packs
 * edge_embedding with rot = 16 so that the positional block comes after the RBF blocks. */
int mpnn_pack_linear_split(const float* W, int out_f, int in_f, int ld, int col0, int rot, void* packed,
                           void* stream);

/* Element-wise probe of the device arithmetic every MLP on the path is built from (used by the unit tests):
 * mode 0: y = GELU(x), the exact-erf form of nn.GELU() (:70, :115, :149) as the kernels evaluate it
 *         (relu(x) - |x| 2^(-Q(|x|)), Q a degree-6 polynomial; within 3e-7 absolute of the fp64 erf form);
 * mode 1: y = hi + lo of the f16 hi/lo operand split that feeds the MFMAs (|x - y| <= 2^-21 |x| for 6e-5 < |x| < 65504);
 * mode 2: y = the single f16 rounding of the *_f16 entry points' operands (round to nearest even, +-inf from
 *         |x| >= 65520 on: the reduced-precision mode overflows loudly too). */
int mpnn_probe_elementwise(const float* x, int n, int mode, float* y, void* stream);

/* OPERAND RANGE of every entry point that takes "packed" weights or hE_frag (all edge / node / featurizer kernels):
 * products run on the f16 matrix pipe, each fp32 operand split as f16 hi + f16 lo.  hi is a truncating conversion
 * (saturates at 65504), lo = f16(x - hi) overflows to +-inf once |x| >= 131008, so an activation or weight beyond that
 * turns the outputs into inf / NaN - never into a silently wrong finite number.  The *_f16 twins drop lo and round hi
 * to nearest even with gfx950's v_cvt_pk_f16_f32, which overflows to +-inf from |x| >= 65520 on: they fail the same
 * loud way (round 2 truncated with v_cvt_pkrtz there, which saturates at 65504 - silently).  Below 6e-5 an operand keeps 3e-8 ABSOLUTE resolution (f16
 * subnormals), i.e. fewer than 22 significant bits.  mpnn_range_flag is the cheap device-side check the Python
 * wrapper uses: weights at pack time against 65504, log_probs after a forward against +inf.
 *   *flag |= bit if any x[i] is NaN or |x[i]| >= limit (limit = INFINITY: finiteness only).  No reference counterpart. */
int mpnn_range_flag(const float* x, long long n, float limit, int32_t* flag, int bit, void* stream);

/* PositionalEncodings as a table: table[d][p] = W[p][d] + bias[p]  (one_hot(d) @ W^T + b, :168-177). */
int mpnn_pos_table(const float* W, const float* bias, int nbuckets /*66*/, int npos /*16*/, float* table,
                   void* stream);

/* X[B,L,4,3] (+ optional noise, :311-315) -> atoms[B*L][6][4] = N,CA,C,O, virtual CB (:317-320),
 * virtual centre V (:215-230); ca_soa[B][4][L] = CA.x, CA.y, CA.z, mask. */
int mpnn_backbone_atoms(const float* X, const float* noise, const float* mask, int B, int L, float* atoms,
                        float* ca_soa, void* stream);

/* ProteinFeatures._dist (:270-289): K nearest by the masked / cutoff-adjusted CA distance, ascending, ties by
 * column index.  edge_cutoff <= 0 means none.  E_idx64 [B,L,K], D_neighbors [B,L,K], edge_mask [B,L,K] may be
 * NULL; E_idx32 [B*L][16*ceil(K/16)] is required. */
int mpnn_knn_topk(const float* ca_soa, int B, int L, int K, float edge_cutoff, int64_t* E_idx64, int32_t* E_idx32,
                  float* D_neighbors, float* edge_mask, void* stream);

/* ProteinFeatures._compute_occupancy (:232-268) on the virtual centres in `atoms`; occ[B*L]. */
int mpnn_occupancy(const float* atoms, const float* mask, int B, int L, float cutoff, float* occ, void* stream);

/* RBF (:291-307, :344-347) + positional (:349-356) features -> edge_embedding + norm_edges (:357-359) -> W_e (:464).
 * n_atoms 5 or 6 (virtual centre).  Wemb_packed from edge_embedding.weight [128, 16+16*n_atoms^2];
 * rbf_mu = the 16 Gaussian centres.  E_natural [B,L,K,128] (the reference's `E`) may be NULL.
 * The RBF distances use the hardware square root (1 ulp; a feature moves by < 2e-6 relative) - only mpnn_knn_topk,
 * whose ranks must equal the reference's, needs the correctly rounded one.  K <= 48 runs as persistent workgroups
 * with the weights staged through LDS, K <= 64 with the weights streamed per residue; same results bit for bit. */
int mpnn_edge_features(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                       const int64_t* chain_labels, const float* pos_table, const float* rbf_mu,
                       const float* Wemb_packed, const float* ln_w, const float* ln_b, const float* We_packed,
                       const float* We_bias, int B, int L, int K, int n_atoms, float* E_natural, float* hE_frag,
                       void* stream);

/* mpnn_edge_features with a bias on the embedding before its LayerNorm: the sparse variant's edge_mlp =
 * Linear(with bias) -> LayerNorm -> Linear (src/mpnn/models/protein_mpnn.py:286-292, :484).  Wemb_bias [128]. */
int mpnn_edge_features_biased(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                              const int64_t* chain_labels, const float* pos_table, const float* rbf_mu,
                              const float* Wemb_packed, const float* Wemb_bias, const float* ln_w, const float* ln_b,
                              const float* We_packed, const float* We_bias, int B, int L, int K, int n_atoms,
                              float* E_natural, float* hE_frag, void* stream);

/* layout converters for API surfaces that expose h_E as [n_nodes, K, 128] (EncLayer / features, :73, :360). */
int mpnn_edges_to_natural(const float* hE_frag, int n_nodes, int K, float* natural, void* stream);
int mpnn_edges_from_natural(const float* natural, int n_nodes, int K, float* hE_frag, void* stream);

/* decoding_order -> rank (inverse permutation): replaces the one_hot + O(L^3) einsum (:492-506). */
int mpnn_order_rank(const int64_t* order, int B, int L, int32_t* rank, void* stream);
/* mask_bw / mask_fw [B,L,K] (:507-519); edge_mask may be NULL. */
int mpnn_order_masks(const int32_t* rank, const int64_t* E_idx64, const float* mask, const float* edge_mask, int B,
                     int L, int K, float* mask_bw, float* mask_fw, void* stream);

/* EncLayer message (:75-81) with W1 split per input block and W3 hoisted:
 *   gsum[i] = sum_k m_ik gelu(W2 gelu(P_i + Q_{j(i,k)} + W1b e_ik) + b2),  cnt[i] = sum_k m_ik,
 *   m_ik = mask_i mask_j edge_mask_ik (:467-470).  P = W1a h + b1 and Q = W1c h are columns
 *   [offP, offP+128) / [offQ, offQ+128) of the node-projection buffer PQ (row stride ldpq; ldpq = 0: one row for
 *   every node - before the first layer h_V = 0 (:448) and P | Q are the bias row, no projection launch needed). */
int mpnn_enc_message(const float* hE_frag, const float* PQ, int ldpq, int offP, int offQ, const int32_t* E_idx32,
                     const float* mask, const float* edge_mask, const float* W1b_packed, const float* W2_packed,
                     const float* b2, int B, int L, int K, float* gsum, float* cnt, void* stream);

/* EncLayer edge update (:90-95), in place on hE_frag:
 *   e_ik <- LN3(e_ik + W13 gelu(W12 gelu(P'_i + Q'_j + W11b e_ik) + b12) + b13). */
int mpnn_enc_edge_update(float* hE_frag, const float* PQ, int ldpq, int offP, int offQ, const int32_t* E_idx32,
                         const float* W11b_packed, const float* W12_packed, const float* b12,
                         const float* W13_packed, const float* b13, const float* ln_w, const float* ln_b, int B,
                         int L, int K, void* stream);

/* Decoder input assembly + DecLayer message (:478-483, :507-525, :121-127), nothing materialised:
 *   s_ik = mask_i edge_mask_ik;  Row_ik = (rank_j < rank_i) ? A_j : PVenc_j   (ONE gathered row per edge)
 *   gsum[i] = sum_{k<K} gelu(W2 gelu(Pa_i + s_ik (Row_ik + W1b e_ik)) + b2)
 *   Pa = W1a h + b1 and A = W1d h + W1c W_s[S] are columns [offPa,+128) / [offA,+128) of PQ (mpnn_node_update
 *   with its table-add); PVenc = W1d h_enc is columns [offEnc,+128) of a buffer with row stride ldenc. */
int mpnn_dec_message(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA, const float* PVenc,
                     int ldenc, int offEnc, const int32_t* rank, const int32_t* E_idx32, const float* mask,
                     const float* edge_mask, const float* W1b_packed, const float* W2_packed, const float* b2, int B,
                     int L, int K, float* gsum, void* stream);

/* mpnn_dec_message for B rows that decode only n_struct distinct structures (row r: structure r % n_struct; K <= 48):
 * hE_frag, E_idx32, edge_mask and the encoder rows PVenc hold n_struct * L residues and are shared by the rows;
 * PQ (Pa | A), rank, mask and gsum hold B * L.  The reference repeats the structure B times and runs featurizer and
 * encoder on every copy (energy_mpnn.py:42-62); SURVEY section 8f N2. */
int mpnn_dec_message_shared(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA, const float* PVenc,
                            int ldenc, int offEnc, const int32_t* rank, const int32_t* E_idx32, const float* mask,
                            const float* edge_mask, const float* W1b_packed, const float* W2_packed, const float* b2,
                            int B, int n_struct, int L, int K, float* gsum, void* stream);

/* Decoder message of the sparse variant (src/mpnn/models/protein_mpnn.py Decoder.message, :203-217, summed over the
 * edges of the target by MessagePassing's default "add"): every edge reads the neighbour's CURRENT projections - the
 * A block (neighbour term + label term) where the neighbour is decoded earlier (rank_i > rank_j), else the B block
 * (neighbour term only) - and only edges with edge_mask = 1 enter the neighbour sum (the dense kernels sum all K).
 * PQ [B*L, ldpq] holds Pa | A | B at offPa / offA / offB; edge_mask [B,L,K] is required. */
int mpnn_dec_message_valid(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA, int offB,
                           const int32_t* rank, const int32_t* E_idx32, const float* mask, const float* edge_mask,
                           const float* W1b_packed, const float* W2_packed, const float* b2, int B, int L, int K,
                           float* gsum, void* stream);

/* One decoder layer of the sampler's level loop, as mpnn_decode_run takes it (HOST struct):
 *   P / RowA (ld, off): the layer's Pa | A rows, per sample row; offEnc: column of its encoder rows in PVenc;
 *   hV_in -> hV_out; PQ [.., nproj * 128]: the NEXT layer's Pa | A rows (nproj = 0 for the last layer);
 *   layers[l].hV_in must be layers[l-1].hV_out and layers[l].P / RowA must live in layers[l-1].PQ.
 * (Round 2 also had mpnn_decode_level, a whole level in one launch; it measured slower than the per-layer launches -
 * profiles/r02_sampler_levels.txt - and was removed in round 3.) */
typedef struct mpnn_level_layer {
  const float* P; int ldp, offP;
  float* RowA; int lda, offA;      /* written too: the drawn letter's term is folded into the A rows */
  int offEnc;
  const float *W1b_packed, *W2_packed, *b2;
  const float* hV_in;
  const float *W3_packed, *b3, *ln1_w, *ln1_b, *Win_packed, *b_in, *Wout8_packed, *b_out, *ln2_w, *ln2_b;
  float scale;
  const float *Wproj_packed, *bproj; int nproj;
  float* hV_out; float* PQ;
} mpnn_level_layer;

/* Node update shared by EncLayer / DecLayer (:81-88, :127-137, FFN :151-155) followed by the next kernels'
 * node projections:  dh = (W3 gsum + cnt b3) / scale;  x = LN1(h + dh);  h' = mask LN2(x + FFN(x));
 *   PQ[n, 128 p : 128 (p+1)] = Wproj_p h' + bproj_p,  p < nproj;
 *   if add_table: PQ[n, block add_block] += add_table[add_idx[n]][off_table : off_table+128]  (row stride ld_table;
 *   the decoder's per-letter sequence term W1c W_s[S_n], :478-479).
 * cnt == NULL uses cnt_const for every node (decoder: K).  do_update = 0 only projects hV_in.
 * do_update = 2 is PositionWiseFeedForward.forward on its own (:151-155): hV_out = W_out gelu(W_in hV_in + b_in) + b_out
 * (only Win_packed / b_in / Wout_packed / b_out / hV_out are read; 128 -> 512 -> 128). */
int mpnn_node_update(const float* hV_in, const float* gsum, const float* cnt, float cnt_const, const float* mask,
                     const float* W3_packed, const float* b3, const float* ln1_w, const float* ln1_b,
                     const float* Win_packed, const float* b_in, const float* Wout_packed, const float* b_out,
                     const float* ln2_w, const float* ln2_b, float scale, int do_update, const float* Wproj_packed,
                     const float* bproj, int nproj, const float* add_table, int ld_table, int off_table,
                     const int64_t* add_idx, int add_block, int n_nodes, float* hV_out, float* PQ, void* stream);

/* Optimizer step of the reference's training loop (train.py:217 torch.optim.AdamW(lr, weight_decay=1e-2), :389-393
 * clip_grad_norm_ then optimizer.step) for ALL parameter tensors in one launch per 48 tensors.
 * params / grads / exp_avg / exp_avg_sq / numel are HOST arrays of n device pointers / element counts (fp32 tensors).
 *   mpnn_grad_sumsq: sumsq[0] = sum of squares of every gradient element (device scalar).  Per-workgroup partials go
 *     to `workspace` and are added in a fixed tree (no atomics: the clip coefficient, hence the update, is the same on
 *     every run); workspace_floats >= sum over groups of 48 tensors of ceil(largest numel / 2048) * tensors in group.
 *   mpnn_grad_scale: grads *= min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))       (torch.nn.utils.clip_grad_norm_)
 *   mpnn_adamw_step: torch.optim.AdamW's update (decoupled weight decay, bias correction for `step` >= 1, no amsgrad);
 *     when grad_sumsq != NULL and max_norm > 0 the clip coefficient above is applied to the gradients on the fly, so
 *     clipping costs no extra pass over them and nothing synchronises with the host. */
int mpnn_grad_sumsq(int n, const float* const* grads, const int* numel, float* sumsq, float* workspace,
                    int workspace_floats, void* stream);
int mpnn_grad_scale(int n, float* const* grads, const int* numel, const float* sumsq, float max_norm, void* stream);
int mpnn_adamw_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int* numel, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, const float* grad_sumsq, float max_norm, void* stream);

/* The gather helpers of the reference as ONE indexed row copy (gather_edges :13-18, gather_nodes :21-30,
 * gather_nodes_t :33-38, cat_neighbors_nodes :41-45):
 *   out[(g * n_idx + q) * out_ld + out_col0 + c] = src[(g * n_src + idx[g * n_idx + q]) * C + c],  c < C, g < G
 * gather_nodes: G = B, n_src = L, n_idx = L*K;  gather_nodes_t: n_idx = K;  gather_edges: G = B*L, n_src = L,
 * n_idx = K;  cat_neighbors_nodes: out_ld = C_nb + C, out_col0 = C_nb (the neighbour features are copied into the
 * leading columns by a second call with idx = NULL, which means idx = q).  C, out_ld, out_col0 multiples of 4;
 * indices must be in [0, n_src) (the Python wrapper checks; out-of-range indices are clamped here, never followed). */
int mpnn_gather_indexed(const float* src, const int64_t* idx, long long G, int n_src, long long n_idx, int C,
                        float* out, int out_ld, int out_col0, void* stream);

/* PositionalEncodings.forward (:168-177): E[e, :] = table[d_e], d = clip(offset + max_rel, 0, 2 max_rel) * mask +
 * (1 - mask) * (2 max_rel + 1), table from mpnn_pos_table (row d = W[:, d] + bias: one_hot(d) @ W^T + b). */
int mpnn_positional_encodings(const int64_t* offset, const int64_t* mask, long long n, int max_rel, const float* table,
                              int npos, float* out, void* stream);

/* y[r, 0:128] = sum_b W_b x[r, 128 b : 128 (b+1)] + bias: nn.Linear(in_f -> 128) over n_rows rows on the matrix pipe,
 * in_f a multiple of 128, W_packed = in_f / 128 consecutive split-packed 128x128 blocks (mpnn_pack_linear_split of
 * W[:, 128 b : 128 (b+1)]), bias may be NULL.  The per-edge term W1[:, 128:] h_ESV of the stand-alone DecLayer
 * (:121-124) when the caller materialises the reference's [B,L,K,384] decoder input. */
int mpnn_linear_rows(const float* x, int ldx, int in_f, const float* W_packed, const float* bias, long long n_rows,
                     float* y, void* stream);
int mpnn_linear_rows_f16(const float* x, int ldx, int in_f, const float* W_packed, const float* bias, long long n_rows,
                         float* y, void* stream);

/* occupancy_mlp (:414-421, :461-462): hV = LN(W2 gelu(w0 occ + b0) + b2). */
int mpnn_occupancy_mlp(const float* occ, const float* w0, const float* b0, const float* W2_packed, const float* b2,
                       const float* ln_w, const float* ln_b, int n_nodes, float* hV, void* stream);

/* W_out + log_softmax (:527-528); n_letters <= 32; log_probs [n_nodes, n_letters]. */
int mpnn_logits_log_softmax(const float* hV, const float* Wout_packed, const float* bias, int n_letters, int n_nodes,
                            float* log_probs, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Training step: per-edge kernels with saved pre-activations and their backward.  The reference differentiates
 * EncLayer.forward (:73-95) / DecLayer.forward (:118-138) with torch autograd inside train.py:375-392; here the
 * same gradients come from explicit kernels.  All *_frag arrays use hE's fragment layout; P / Row* / dP / out*
 * are dense [n_nodes,128]; *_rows are row-major [n_nodes * 16 ceil(K/16)][128].  K <= 64 (K > 48 runs the
 * four-column-tile instances: correct, register-spilling, untuned).
 * ------------------------------------------------------------------------------------------------------------ */

/* Forward of mpnn_enc_message (dec = 0: Row = RowA[j], cnt written) or mpnn_dec_message (dec = 1: Row = RowA[j]
 * where rank[j] < rank[i], else RowB[j]) that also stores a1 = P + s (Row + W1b e) and a2 = W2 gelu(a1) + b2.
 * w_valid != 0 (dec = 1 only, edge_mask required): the neighbour sum runs over the edges with edge_mask = 1 only, as
 * mpnn_dec_message_valid's does (the edge-list model's Decoder, src/mpnn/models/protein_mpnn.py:203-217). */
int mpnn_message_train(int dec, const float* hE_frag, const float* P, const float* RowA, const float* RowB,
                       const int32_t* rank, const int32_t* E_idx32, const float* mask, const float* edge_mask,
                       const float* W1b_packed, const float* W2_packed, const float* b2, int B, int L, int K,
                       float* gsum, float* cnt, float* a1_frag, float* a2_frag, int w_valid, void* stream);

/* Forward of mpnn_enc_edge_update, out of place, storing a1', a2' and the pre-LayerNorm sum z (:90-95).
 * p_drop > 0 applies EncLayer.dropout3 (:94) to the MLP output: an element is kept when
 * mix32(element_index * 0x9E3779B1 ^ drop_seed) >= p_drop * 2^32 (element_index = its float index in the fragment
 * array) and scaled by 1 / (1 - p_drop); the backward call must be given the same p_drop and drop_seed.
 * rstd_max (one float, zero it first; may be NULL): receives the largest 1 / sqrt(var + eps) of the launch's
 * LayerNorms, which mpnn_enc_edge_update_bwd_fused needs for its operand scales. */
int mpnn_enc_edge_update_train(const float* hE_frag, const float* P, const float* Q, const int32_t* E_idx32,
                               const float* W11b_packed, const float* W12_packed, const float* b12,
                               const float* W13_packed, const float* b13, const float* ln_w, const float* ln_b, int B,
                               int L, int K, float* hE_out, float* a1_frag, float* a2_frag, float* z_frag,
                               float p_drop, uint32_t drop_seed, float* rstd_max, void* stream);

/* Backward of the message MLP: given dgsum, writes da2 (fragment), s-scaled da1 rows, dP = sum_k da1, and
 * dE (+)= W1b^T (s da1).  W2T / W1bT are mpnn_pack_linear_split of the TRANSPOSED matrices.
 * amax2[0..1] (zero them first) receive max |da2| and max |da1 rows| as fp32: mpnn_edge_wgrad's A_absmax.
 * w_valid as given to mpnn_message_train. */
int mpnn_message_bwd(int dec, const float* dgsum, const float* a1_frag, const float* a2_frag, const int32_t* E_idx32,
                     const float* mask, const float* edge_mask, const float* W2T_packed, const float* W1bT_packed,
                     int B, int L, int K, float* da2_frag, float* da1_rows, float* dP, float* dE_frag, int accumulate,
                     float* amax2, int w_valid, void* stream);

/* mpnn_message_bwd AND its two weight-gradient contractions (dW2 += da2 (x) gelu(a1), db2 += sum da2,
 * dW1b += rows (x) e) in one kernel, for K <= 48 (above: mpnn_message_bwd + 2 x mpnn_edge_wgrad): da2 is never written,
 * a1 / hE / the rows are read once instead of twice (reference: the autograd of model_utils.py:241-243 / :112-115,
 * whose eager form materialises every one of these tensors).  dW2 / db2 / dW1b are ACCUMULATED INTO (zero them for a
 * plain gradient).  The f16 hi/lo operand splits use launch-wide power-of-two scales from bounds (1.13 max |dgsum|,
 * times max_k sum_o |W2[o][k]| for the rows), found by a small pre-launch: no host value is needed.
 * dE_absmax (one float, zero it before the FIRST writer of a dE buffer; may be NULL): atomic max of |dE| as written.
 * workspace: 64 + 33024 floats per workgroup, one workgroup per CU at most (mpnn_workspace_bytes); partial sums are
 * added in a fixed order - bit-identical from run to run. */
int mpnn_message_bwd_fused(int dec, const float* dgsum, const float* a1_frag, const float* a2_frag,
                           const float* hE_frag, const int32_t* E_idx32, const float* mask, const float* edge_mask,
                           const float* W2T_packed, const float* W1bT_packed, int B, int L, int K, float* da1_rows,
                           float* dP, float* dE_frag, int accumulate, int w_valid, float* dW2, float* db2, float* dW1b,
                           float* dE_absmax, float* workspace, int workspace_floats, void* stream);

/* mpnn_enc_edge_update_bwd AND the weight-gradient contractions of W13 and W12 (dW13 += dzd (x) gelu(a2),
 * db13 += sum dzd, dW12 += da2 (x) gelu(a1), db12 += sum da2) in one kernel, for K <= 48 (above:
 * mpnn_enc_edge_update_bwd + 3 x mpnn_edge_wgrad): dz and da2 are never written, a2 / a1 are read once (reference: the
 * autograd of EncLayer's edge update, model_utils.py:90-95, whose eager form materialises every one of these tensors).
 * What remains for the caller: dW11b = mpnn_edge_wgrad(da1_rows, hE) with da1_absmax as its A_absmax.
 * dE_frag must NOT alias dy_frag.  dW13 / db13 / dW12 / db12 / dln are ACCUMULATED INTO (zero them for a plain gradient).
 * Operand scales (f16 hi/lo splits of gradient-sized values) are launch-wide powers of two from BOUNDS:
 *   |dz| <= 13.3 * rstd_max * max |ln_w| * dy_absmax / (1 - p_drop), then 1.13 * max_k sum_o |W[o][k]| per GEMM;
 * dy_absmax (one float on the device: max |dy| or any upper bound of it - the dE_absmax output of the kernels that
 * wrote dy) and rstd_max (mpnn_enc_edge_update_train's output) are read by a small pre-launch: no host value is needed.
 * da1_absmax, dE_absmax (one float each, zero them first; dE_absmax may be NULL) receive max |da1 rows| and max |dE|.
 * workspace: 64 + 33280 floats per workgroup, one workgroup per CU at most (mpnn_workspace_bytes); partial sums are
 * added in a fixed order - bit-identical from run to run. */
int mpnn_enc_edge_update_bwd_fused(const float* dy_frag, const float* z_frag, const float* a2_frag,
                                   const float* a1_frag, const float* ln_w, const float* W13T_packed,
                                   const float* W12T_packed, const float* W11bT_packed, int B, int L, int K,
                                   float* da1_rows, float* da1_absmax, float* dP, float* dE_frag, float* dln,
                                   float* dW13, float* db13, float* dW12, float* db12, const float* dy_absmax,
                                   const float* rstd_max, float* dE_absmax, float p_drop, uint32_t drop_seed,
                                   float* workspace, int workspace_floats, void* stream);

/* Backward of the edge update: LayerNorm', W13^T, W12^T, W11b^T; dln[0:128] += d ln_w, dln[128:256] += d ln_b.
 * dE_frag may alias dy_frag.  amax3[0..2] (zero them first) receive max |dz|, |da2|, |da1 rows|.
 * workspace: >= 256 floats per workgroup (one workgroup per CU at most: 65536 floats always suffice) for the
 * LayerNorm-parameter partial sums, which a second small launch adds in a fixed order - no floating-point atomics
 * anywhere in the backward pass, every gradient is bit-identical from run to run. */
int mpnn_enc_edge_update_bwd(const float* dy_frag, const float* z_frag, const float* a2_frag, const float* a1_frag,
                             const float* ln_w, const float* W13T_packed, const float* W12T_packed,
                             const float* W11bT_packed, int B, int L, int K, float* dz_frag, float* da2_frag,
                             float* da1_rows, float* dP, float* dE_frag, float* dln, float* amax3, float p_drop,
                             uint32_t drop_seed, float* workspace, int workspace_floats, void* stream);

/* dW[fo][fi] += sum_edges A[e][fo] * f(B[e][fi]), dbias[fo] += sum_edges A[e][fo]  (the nn.Linear weight / bias
 * gradients of one per-edge 128x128 layer).  A: fragment layout, or rows when a_rows != 0; B: fragment layout,
 * f = GELU when b_gelu != 0.  dW [128,128] row-major, accumulated with atomics (zero it first); dbias may be NULL.
 * A_absmax: device scalar >= max |A| (any positive value of that magnitude): gradients are tiny, so A is scaled by a
 * power of two derived from it before the f16 hi/lo split and the result scaled back.
 * workspace: device scratch for per-workgroup partial sums, 16512 floats per workgroup; 2 x CUs x 16512 floats (what
 * mpnn_workspace_bytes reports, with room to spare) lets the
 * launch fill the chip (fewer -> fewer workgroups). */
int mpnn_edge_wgrad(const float* A, int a_rows, const float* A_absmax, const float* B_frag, int b_gelu, int n_nodes,
                    int K, float* dW, float* dbias, float* workspace, int workspace_floats, void* stream);

/* Gradient of the neighbour gather h[E_idx]: outA[j] = sum of rows[inv_row[m]] for m in [inv_ptr[j], inv_ptr[j+1]),
 * split between outA (sel[m] == 0) and outB (sel[m] != 0) when sel is given (decoder: which block the edge read).
 * Work unit = a chunk of <= 64 list entries: node j owns max(1, ceil(degree_j / 64)) chunks, chunk_ptr [n_nodes + 1]
 * their offsets, chunk_node[c] the node of chunk c, max_chunks >= chunk_ptr[n_nodes] (n_edges / 64 + n_nodes always
 * suffices).  Every output row is WRITTEN (no zero fill needed), in list order with a fixed summation tree: no atomics,
 * bit-identical from run to run.  partials: device scratch of max_chunks * 256 floats (the partial sums of the nodes
 * that own several chunks, added in chunk order by a second small launch). */
int mpnn_gather_rows(const float* rows, const int32_t* inv_ptr, const int32_t* inv_row, const unsigned char* sel,
                     const int32_t* chunk_ptr, const int32_t* chunk_node, int n_nodes, int max_chunks, float* outA,
                     float* outB, float* partials, void* stream);

/* Inverse neighbour lists for mpnn_gather_rows (the index side of the autograd of gather_nodes, :19-26, which the
 * reference leaves to torch's scatter-add): inv_row [B*L*K] = edge ids (i * Kp + k, Kp = 16 ceil(K/16)) grouped by the
 * node j = E_idx[i][k] they gather from and ordered by (i, k) inside a group - the order of a stable sort, the same
 * on every run; inv_ptr [B*L + 1] the group offsets, chunk_ptr [B*L + 1] / chunk_node [B*L*K / 64 + B*L] the chunk
 * table of mpnn_gather_rows.
 * E_idx32 [B*L, Kp] as mpnn_knn_topk writes it (ids local to the protein, a residue's K neighbours distinct).  The
 * construction keeps ONE edge per (target, centre) pair: a residue that lists a neighbour twice, or an id outside
 * [0, L), loses an edge - status (optional device word) is then set to 1, else 0; callers with lists of unknown origin
 * check it and fall back to a sort (mpnn_amd.training.Graph does).
 * slot_map: B*L*L bytes of device scratch.  1 <= K <= min(64, L). */
int mpnn_inverse_lists(const int32_t* E_idx32, int B, int L, int K, unsigned char* slot_map, int32_t* inv_ptr,
                       int32_t* chunk_ptr, int32_t* chunk_node, int32_t* inv_row, int32_t* status, void* stream);
/* sel [B*L*K] for mpnn_gather_rows in the decoder (:507-525): 1 where the gathering residue i does not come after its
 * neighbour j in the decoding order (rank[i] <= rank[j]: the edge read the encoder block), in list order. */
int mpnn_inverse_sel(const int32_t* rank, const int32_t* inv_ptr, const int32_t* inv_row, int n_nodes, int K,
                     unsigned char* sel, void* stream);

/* Training forward of mpnn_edge_features: also stores the pre-LayerNorm embedding E0 and E = norm_edges(E0), both in
 * the fragment layout, for mpnn_edge_features_bwd / mpnn_edge_wgrad (:334-358, :464).  Wemb_bias [128] or NULL: a
 * bias on the embedding before its LayerNorm (mpnn_edge_features_biased's form, the edge-list model's edge_mlp.0);
 * its gradient is the column sum of dE0, i.e. mpnn_edge_wgrad's dbias output. */
int mpnn_edge_features_train(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                             const int64_t* chain_labels, const float* pos_table, const float* rbf_mu,
                             const float* Wemb_packed, const float* Wemb_bias, const float* ln_w, const float* ln_b,
                             const float* We_packed, const float* We_bias, int B, int L, int K, int n_atoms,
                             float* E0_frag, float* E_frag, float* hE_frag, void* stream);

/* Featurizer backward: dE = W_e^T dhE0, dE0 = LayerNorm'(dE; E0).  WeT_packed = mpnn_pack_linear_split(W_e^T).
 * dln[0:128] += d norm_edges.weight, dln[128:256] += d norm_edges.bias; amax2 (zeroed) <- max |dhE0|, max |dE0|.
 * workspace as for mpnn_enc_edge_update_bwd (256 floats per workgroup). */
int mpnn_edge_features_bwd(const float* dhE_frag, const float* E0_frag, const float* ln_w, const float* WeT_packed,
                           int B, int L, int K, float* dE0_frag, float* dln, float* amax2, float* workspace,
                           int workspace_floats, void* stream);

/* The edge_embedding layer's raw inputs in the fragment layout, 128 features per chunk, chunks back to back:
 * feature 16 p + r (p < n_atoms^2) = RBF r of atom pair p (:334-347); features 16 n_atoms^2 + d = one-hot of the
 * relative-position bucket d in [0, 66) (:349-356); zero padding up to ceil((n_atoms^2 + 5) / 8) chunks.
 * Contracted with dE0 by mpnn_edge_wgrad they give the edge_embedding / positional-linear weight gradients. */
int mpnn_feature_chunks(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                        const int64_t* chain_labels, const float* rbf_mu, int B, int L, int K, int n_atoms,
                        float* chunks_frag, void* stream);

/* The weight gradient those chunks exist for, without materialising them (training backward of edge_embedding,
 * model_utils.py:334-358): dF[c][o][i] += sum over residues and edges of dE0[.., o] * F_c[.., i], c = 0 .. nch - 1
 * (nch = ceil((n_atoms^2 + 5) / 8) chunks of 128 raw features: the 16 RBFs of each atom pair, then the one-hot of the
 * 66 relative-position buckets, zero beyond), each chunk's values computed inside the contraction kernel from atoms,
 * neighbour ids, residue numbers and chain labels exactly as mpnn_edge_features_train computes them.  dE0_frag: the
 * gradient of the pre-LayerNorm embedding in the fragment layout, dE0_absmax its largest magnitude (device scalar, as
 * mpnn_edge_features_bwd leaves it).  dF [nch][128][128] accumulates: zero it first.  dbias [128] (may be NULL)
 * accumulates the column sums of dE0 (the edge-list variant's embedding bias).  Deterministic (ordered partials);
 * workspace as for mpnn_edge_wgrad. */
int mpnn_feature_wgrad(const float* dE0_frag, const float* dE0_absmax, const float* atoms, const int32_t* E_idx32,
                       const int64_t* residue_idx, const int64_t* chain_labels, const float* rbf_mu, int B, int L, int K,
                       int n_atoms, float* dF, float* dbias, float* workspace, int workspace_floats, void* stream);

/* Training forward of mpnn_node_update (:81-88, :127-137, :151-155): also stores x1pre = h + drop(dh), the FFN
 * pre-activation a [n_nodes,512] and x2pre = x1 + drop(FFN) for the backward kernel; applies the layers' two node-level
 * dropouts (counter-based, ids node*128+feature; the second with seed ^ 0x5bd1e995) when p_drop > 0; writes the
 * projections block-major PQ_blocks[p][node][128].  No table add (the sequence term is added by the caller). */
int mpnn_node_update_train(const float* hV_in, const float* gsum, const float* cnt, float cnt_const, const float* mask,
                           const float* W3_packed, const float* b3, const float* ln1_w, const float* ln1_b,
                           const float* Win_packed, const float* b_in, const float* Wout_packed, const float* b_out,
                           const float* ln2_w, const float* ln2_b, float scale, const float* Wproj_packed,
                           const float* bproj, int nproj, int n_nodes, float* hV_out, float* PQ_blocks, float* x1pre,
                           float* a_pre, float* x2pre, float p_drop, uint32_t drop_seed, void* stream);

/* Backward of the node update.  Inputs: dY = d hV_out (may be NULL), dPQ_blocks[p][node][128], the saved tensors.
 * *T_packed = mpnn_pack_linear_split of the transposed matrices (WoutT: the four 128-column blocks of W_out transposed,
 * WinT: the four 128-row blocks of W_in transposed, consecutive).  Outputs: d_hV_in, d_gsum, and the operands of the
 * weight gradients (mpnn_node_wgrad): x1 [n,128], hid = gelu(a) [n,512], d_a [n,512], d_o [n,128], d_dh [n,128];
 * dvec640 (zero it first) += d ln1_w | d ln1_b | d ln2_w | d ln2_b | d b3 (per-workgroup partials in `workspace`,
 * >= 640 floats per 64 nodes, added in a fixed order by a second small launch). */
int mpnn_node_update_bwd(const float* dY, const float* dPQ_blocks, const float* x1pre, const float* a_pre,
                         const float* x2pre, const float* cnt, float cnt_const, const float* mask, const float* ln1_w,
                         const float* ln1_b, const float* ln2_w, float scale, const float* WprojT_packed, int nproj,
                         const float* WoutT_packed, const float* WinT_packed, const float* W3T_packed, int n_nodes,
                         float* d_hV_in, float* d_gsum, float* x1, float* hid, float* d_a, float* d_o, float* d_dh,
                         float* dvec640, float p_drop, uint32_t drop_seed, float* workspace, int workspace_floats,
                         void* stream);

/* mpnn_node_update_bwd with the projections' gradients given one by one: dPQ_each is a HOST array of nproj device
 * pointers, entry p = d PQ_p [n][128] wherever its producer left it, NULL = that projection received no gradient
 * (autograd delivers them as separate tensors; stacking them for dPQ_blocks copies nproj x n x 128 floats per layer).
 * Everything else as above. */
int mpnn_node_update_bwd_each(const float* dY, const float* const* dPQ_each, const float* x1pre, const float* a_pre,
                              const float* x2pre, const float* cnt, float cnt_const, const float* mask,
                              const float* ln1_w, const float* ln1_b, const float* ln2_w, float scale,
                              const float* WprojT_packed, int nproj, const float* WoutT_packed,
                              const float* WinT_packed, const float* W3T_packed, int n_nodes, float* d_hV_in,
                              float* d_gsum, float* x1, float* hid, float* d_a, float* d_o, float* d_dh, float* dvec640,
                              float p_drop, uint32_t drop_seed, float* workspace, int workspace_floats, void* stream);

/* mpnn_pack_linear_split for n 128x128 blocks in one launch per 16 blocks: block b is read at blocks[b] with row
 * stride ld[b] (out[o][i] = W[o * ld + i]), or transposed (out[o][i] = W[i * ld + o]) when transposed[b] != 0, and
 * written to packed + b * 16384 floats.  blocks / ld / transposed are HOST arrays of device pointers / ints. */
int mpnn_pack_blocks(const float* const* blocks, const int* ld, const int* transposed, int n, float* packed,
                     void* stream);

/* The decoder's sequence term in training (model_utils.py:478-483: h_S = W_s(S) is concatenated into the decoder
 * input, i.e. enters the third 128-column block W1c of the layer's W1 [128, 512]).  With TS = W_s W1c^T
 * ([n_letters, 128]) the term of residue n is the row TS[S[n]]:
 *   mpnn_seq_table      TS[l][o] = sum_i W_s[l][i] W1c[o * ld_w1c + i]        (W_s [n_letters, 128] row-major)
 *   mpnn_add_table_rows A[n][:] = A0[n][:] + TS[S[n]][:]                        (rows of 128; S outside [0, n_letters):
 *                                                                                nothing added)
 *   mpnn_seq_table_bwd  dW_s[l][i] = sum_o dTS[l][o] W1c[o][i],  dW1c[o][i] = sum_l dTS[l][o] W_s[l][i]   (both
 *                       overwritten, row-major 128-wide), dTS given TRANSPOSED as dTS_t[o * ld_dts + l] - the form
 *                       mpnn_node_wgrad leaves when it contracts the rows' gradient with one-hot rows of S.
 * Plain fp32 FMAs in a fixed order (gradient-sized operands: no f16 splitting, no scaling, bit-reproducible). */
int mpnn_seq_table(const float* Ws, const float* W1c, int ld_w1c, int n_letters, float* TS, void* stream);
int mpnn_add_table_rows(const float* A0, const float* TS, const int64_t* S, int n_letters, int n_rows, float* A,
                        void* stream);
int mpnn_seq_table_bwd(const float* dTS_t, int ld_dts, const float* Ws, const float* W1c, int ld_w1c, int n_letters,
                       float* dWs, float* dW1c, void* stream);

/* Weight / bias gradient of an nn.Linear over per-residue rows (autograd of the node-level linears, :78-93, :151-155):
 * dW[o][i] += sum_n dY[n][o] X[n][i]  (row strides ldy, ldx, ldw), dbias[o] += sum_n dY[n][o].  Products as three
 * f16-split MFMA terms with fp32 accumulation (relative error per product <= ~2^-21); each 32-row batch of dY is scaled
 * by a power of two taken from its largest magnitude, so gradients of any size keep that precision; |X| < 65504.
 * Accumulates: zero dW / dbias first.  dbias may be NULL.  workspace: device scratch, 16512 floats per 128x128 output
 * block and row chunk (256 chunks per block fill the chip). */
int mpnn_node_wgrad(const float* dY, int ldy, int out_f, const float* X, int ldx, int in_f, int n_rows, float* dW,
                    int ldw, float* dbias, float* workspace, int workspace_floats, void* stream);

/* mpnn_node_wgrad without a workspace argument and WITHOUT ANY ALLOCATION.  No counterpart in the reference: it exists
 * for callers that cannot allocate device memory - code under HIP-graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same arithmetic and kernels as mpnn_node_wgrad; the split-K partials go to a device array built
 * into the library (128 partials x 16512 floats = 8.06 MiB per GPU, held until the process ends).
 * Limits: ceil(out_f/128) * ceil(in_f/128) <= 128 output blocks and n_rows <= 2^24, else -1 with a message (use
 * mpnn_node_wgrad).  The first call on a device looks up that device's copy of the array (make it outside a capture).
 * All calls on one GPU share the array: calls that overlap on one GPU (two streams, two threads) race - serialise them.
 * Accumulates like mpnn_node_wgrad: zero dW / dbias first. */
int mpnn_node_wgrad_noalloc(const float* dY, int ldy, int out_f, const float* X, int ldx, int in_f, int n_rows,
                            float* dW, int ldw, float* dbias, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Autoregressive sampling (BASELINE configs[4]).  The reference has no sampler; the contract is
 * that a teacher-forced forward (:427-529) on the sampled sequence with the same decoding order returns, at every
 * residue, the log-probs the sampler drew from.  Samples of one structure share the encoder output: encoder-side
 * arrays hold enc_nodes = P * L residues, sample-side arrays n_samples * enc_nodes, residue id = s * enc_nodes + n.
 * Residues are processed in dependency levels (a residue's level = 1 + max level of its already-decoded
 * neighbours), each level being one launch per kernel over node_list.
 * ------------------------------------------------------------------------------------------------------------ */

/* Dependency levels of a decoding order: level[row][i] = 1 + max level of i's neighbours decoded before i (1 when
 * none).  order / rank / level are [n_rows][L] (n_rows = n_samples * enc_rows); E_idx32 is the shared [enc_rows*L][16
 * ceil(K/16)] neighbour list, row r uses structure r % enc_rows. */
int mpnn_decode_levels(const int64_t* order, const int32_t* rank, const int32_t* E_idx32, int n_rows, int enc_rows,
                       int L, int K, int32_t* level, void* stream);

/* mpnn_dec_message for the residues in node_list: hE / E_idx32 / edge_mask / PVenc are indexed with
 * (residue % enc_nodes); P, RowA, rank, mask, gsum with the residue id.  One wave per (residue, 16-neighbour tile);
 * gsum is [residue][4][128]: tile c of a residue stores its sums to gsum[residue][c] (plain stores, no atomics;
 * mpnn_node_update_subset adds the ceil(K/16) tiles in order). */
int mpnn_dec_message_subset(const float* hE_frag, const float* P, int ldp, int offP, const float* RowA, int lda,
                            int offA, const float* PVenc, int ldenc, int offEnc, const int32_t* rank,
                            const int32_t* E_idx32, const float* mask, const float* edge_mask, const float* W1b_packed,
                            const float* W2_packed, const float* b2, const int32_t* node_list, int n_list,
                            int enc_nodes, int L, int K, float* gsum, void* stream);

/* One decoder layer for the residues in node_list in a single launch: mpnn_dec_message_subset followed by
 * mpnn_node_update_subset, with the neighbour sums kept in LDS (no gsum buffer).  A workgroup owns 4 listed residues.
 * Wout_packed here is W_out as EIGHT 64-column blocks, each mpnn_pack_linear_split(W_out, col0 = 64 q, ncols = 64). */
int mpnn_dec_layer_subset(const float* hE_frag, const float* P, int ldp, int offP, const float* RowA, int lda, int offA,
                          const float* PVenc, int ldenc, int offEnc, const int32_t* rank, const int32_t* E_idx32,
                          const float* mask, const float* edge_mask, const float* W1b_packed, const float* W2_packed,
                          const float* b2, const int32_t* node_list, int n_list, int enc_nodes, int L, int K,
                          const float* hV_in, float cnt_const, const float* W3_packed, const float* b3,
                          const float* ln1_w, const float* ln1_b, const float* Win_packed, const float* b_in,
                          const float* Wout_packed, const float* b_out, const float* ln2_w, const float* ln2_b,
                          float scale, const float* Wproj_packed, const float* bproj, int nproj, float* hV_out,
                          float* PQ, void* stream);

/* The LAST decoder layer of a level and the draw in one launch: mpnn_dec_layer_subset (without projections: no layer
 * follows) and then mpnn_sample_subset on the same list - the workgroup that finished a residue's node features draws
 * its letter and folds the sequence term into the layers' gathered rows.  Results identical to the two calls
 * (model_utils.py:507-529; sampling has no counterpart in the reference, see mpnn_sample_subset).  K <= 48. */
int mpnn_dec_layer_sample_subset(const float* hE_frag, const float* P, int ldp, int offP, const float* RowA, int lda,
                                 int offA, const float* PVenc, int ldenc, int offEnc, const int32_t* rank,
                                 const int32_t* E_idx32, const float* mask, const float* edge_mask,
                                 const float* W1b_packed, const float* W2_packed, const float* b2,
                                 const int32_t* node_list, int n_list, int enc_nodes, int L, int K, const float* hV_in,
                                 float cnt_const, const float* W3_packed, const float* b3, const float* ln1_w,
                                 const float* ln1_b, const float* Win_packed, const float* b_in,
                                 const float* Wout_packed, const float* b_out, const float* ln2_w, const float* ln2_b,
                                 float scale, float* hV_out, const float* Wlogit_packed, const float* blogit,
                                 const float* logit_bias, int n_letters, float temperature, uint64_t seed,
                                 const float* designable, const int64_t* S_in, int64_t* S_out, float* log_probs,
                                 const float* TS, int ld_ts, int n_layers, float* const* A_rows, const int* A_ld,
                                 const int* A_off, void* stream);

/* mpnn_node_update (do_update = 1, cnt = cnt_const, no table add) for the residues in node_list; gsum [residue][4][128]
 * as mpnn_dec_message_subset writes it, n_tiles = ceil(K/16) of its tiles are added, in tile order. */
int mpnn_node_update_subset(const float* hV_in, const float* gsum, int n_tiles, float cnt_const, const float* mask,
                            const float* W3_packed, const float* b3, const float* ln1_w, const float* ln1_b,
                            const float* Win_packed, const float* b_in, const float* Wout_packed, const float* b_out,
                            const float* ln2_w, const float* ln2_b, float scale, const float* Wproj_packed,
                            const float* bproj, int nproj, const int32_t* node_list, int n_list, float* hV_out,
                            float* PQ, void* stream);

/* Output head + draw for the residues in node_list: log_probs = log_softmax(W_out h + b) (:527-528);
 * S ~ softmax((logits + logit_bias) / temperature) where designable[n] != 0 (uniform variate = splitmix64(seed,
 * residue id)), else S = S_in[n]; then A_rows[l][n * A_ld[l] + A_off[l] ..+128] += TS[S][128 l ..] for every decoder
 * layer l < n_layers <= 4 (the sequence term W1c W_s[S], :478-479).  A_rows / A_ld / A_off are HOST arrays. */
int mpnn_sample_subset(const float* hV, const float* Wout_packed, const float* bias, const float* logit_bias,
                       int n_letters, const int32_t* node_list, int n_list, float temperature, uint64_t seed,
                       const float* designable, const int64_t* S_in, int64_t* S_out, float* log_probs,
                       const float* TS, int ld_ts, int n_layers, float* const* A_rows, const int* A_ld,
                       const int* A_off, void* stream);

/* The whole decoder of a sampling call in ONE call: walks the levels (node_list = the residue ids sorted by level,
 * DEVICE; level_counts [n_levels] = residues per level, HOST) and enqueues, per level, mpnn_dec_layer_subset for every
 * decoder layer but the last and mpnn_dec_layer_sample_subset for the last (draw_fused = 1), or mpnn_dec_layer_subset
 * for all layers followed by mpnn_sample_subset (draw_fused = 0) - the same launches, in the same order, with the same
 * results as those per-level calls, issued from C instead of ~4 ctypes calls per level (the host side of a Python
 * loop over ~70 levels x 3 launches was 3 of the 7.7 ms of a 64 x 512 sampling call).  `layers` is a HOST array of
 * n_layers mpnn_level_layer (layers[n_layers-1].nproj must be 0); the A rows the draw folds
 * the sequence term into are layers[l].RowA / lda / offA.  Returns at the first failing launch.  K <= 48.
 * No reference counterpart (the reference has no sampler). */
int mpnn_decode_run(const float* hE_frag, const float* PVenc, int ldenc, const int32_t* rank, const int32_t* E_idx32,
                    const float* mask, const float* edge_mask, const int32_t* node_list, const int32_t* level_counts,
                    int n_levels, int enc_nodes, int L, int K, float cnt_const, const mpnn_level_layer* layers,
                    int n_layers, const float* Wlogit_packed, const float* logit_b, const float* logit_bias,
                    int n_letters, float temperature, uint64_t seed, const float* designable, const int64_t* S_in,
                    int64_t* S_out, float* log_probs, const float* TS, int ld_ts, int draw_fused, void* stream);
int mpnn_decode_run_f16(const float* hE_frag, const float* PVenc, int ldenc, const int32_t* rank,
                        const int32_t* E_idx32, const float* mask, const float* edge_mask, const int32_t* node_list,
                        const int32_t* level_counts, int n_levels, int enc_nodes, int L, int K, float cnt_const,
                        const mpnn_level_layer* layers, int n_layers, const float* Wlogit_packed, const float* logit_b,
                        const float* logit_bias, int n_letters, float temperature, uint64_t seed,
                        const float* designable, const int64_t* S_in, int64_t* S_out, float* log_probs, const float* TS,
                        int ld_ts, int draw_fused, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Reduced-precision forward ("f16 products"; BASELINE configs[3] asks for bf16-class message MLPs): the same
 * kernels compiled with ONE f16 MFMA product per fp32 product (operands rounded to f16, 11 significant bits - three
 * more than bf16) instead of three; accumulation, GELU, LayerNorm, softmax stay fp32, neighbour search and masks are
 * the exact kernels.  Same arguments and packed-weight buffers as the entry points they shadow.
 * Stated tolerance (tests/test_gpu_parity.py): max |d log_prob| <= 5e-2, argmax agreement >= 99 %.
 * ------------------------------------------------------------------------------------------------------------ */
int mpnn_edge_features_f16(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                       const int64_t* chain_labels, const float* pos_table, const float* rbf_mu,
                       const float* Wemb_packed, const float* ln_w, const float* ln_b, const float* We_packed,
                       const float* We_bias, int B, int L, int K, int n_atoms, float* E_natural, float* hE_frag,
                       void* stream);

int mpnn_edge_features_biased_f16(const float* atoms, const int32_t* E_idx32, const int64_t* residue_idx,
                                  const int64_t* chain_labels, const float* pos_table, const float* rbf_mu,
                                  const float* Wemb_packed, const float* Wemb_bias, const float* ln_w,
                                  const float* ln_b, const float* We_packed, const float* We_bias, int B, int L, int K,
                                  int n_atoms, float* E_natural, float* hE_frag, void* stream);
int mpnn_dec_message_valid_f16(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA, int offB,
                               const int32_t* rank, const int32_t* E_idx32, const float* mask, const float* edge_mask,
                               const float* W1b_packed, const float* W2_packed, const float* b2, int B, int L, int K,
                               float* gsum, void* stream);

int mpnn_enc_message_f16(const float* hE_frag, const float* PQ, int ldpq, int offP, int offQ, const int32_t* E_idx32,
                     const float* mask, const float* edge_mask, const float* W1b_packed, const float* W2_packed,
                     const float* b2, int B, int L, int K, float* gsum, float* cnt, void* stream);

int mpnn_enc_edge_update_f16(float* hE_frag, const float* PQ, int ldpq, int offP, int offQ, const int32_t* E_idx32,
                         const float* W11b_packed, const float* W12_packed, const float* b12,
                         const float* W13_packed, const float* b13, const float* ln_w, const float* ln_b, int B,
                         int L, int K, void* stream);

int mpnn_dec_message_f16(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA, const float* PVenc,
                     int ldenc, int offEnc, const int32_t* rank, const int32_t* E_idx32, const float* mask,
                     const float* edge_mask, const float* W1b_packed, const float* W2_packed, const float* b2, int B,
                     int L, int K, float* gsum, void* stream);
int mpnn_dec_message_shared_f16(const float* hE_frag, const float* PQ, int ldpq, int offPa, int offA, const float* PVenc,
                                int ldenc, int offEnc, const int32_t* rank, const int32_t* E_idx32, const float* mask,
                                const float* edge_mask, const float* W1b_packed, const float* W2_packed, const float* b2,
                                int B, int n_struct, int L, int K, float* gsum, void* stream);

int mpnn_node_update_f16(const float* hV_in, const float* gsum, const float* cnt, float cnt_const, const float* mask,
                     const float* W3_packed, const float* b3, const float* ln1_w, const float* ln1_b,
                     const float* Win_packed, const float* b_in, const float* Wout_packed, const float* b_out,
                     const float* ln2_w, const float* ln2_b, float scale, int do_update, const float* Wproj_packed,
                     const float* bproj, int nproj, const float* add_table, int ld_table, int off_table,
                     const int64_t* add_idx, int add_block, int n_nodes, float* hV_out, float* PQ, void* stream);

int mpnn_dec_message_subset_f16(const float* hE_frag, const float* P, int ldp, int offP, const float* RowA, int lda,
                            int offA, const float* PVenc, int ldenc, int offEnc, const int32_t* rank,
                            const int32_t* E_idx32, const float* mask, const float* edge_mask, const float* W1b_packed,
                            const float* W2_packed, const float* b2, const int32_t* node_list, int n_list,
                            int enc_nodes, int L, int K, float* gsum, void* stream);
int mpnn_dec_layer_subset_f16(const float* hE_frag, const float* P, int ldp, int offP, const float* RowA, int lda,
                              int offA, const float* PVenc, int ldenc, int offEnc, const int32_t* rank,
                              const int32_t* E_idx32, const float* mask, const float* edge_mask,
                              const float* W1b_packed, const float* W2_packed, const float* b2,
                              const int32_t* node_list, int n_list, int enc_nodes, int L, int K, const float* hV_in,
                              float cnt_const, const float* W3_packed, const float* b3, const float* ln1_w,
                              const float* ln1_b, const float* Win_packed, const float* b_in,
                              const float* Wout_packed, const float* b_out, const float* ln2_w, const float* ln2_b,
                              float scale, const float* Wproj_packed, const float* bproj, int nproj, float* hV_out,
                              float* PQ, void* stream);
int mpnn_dec_layer_sample_subset_f16(const float* hE_frag, const float* P, int ldp, int offP, const float* RowA, int lda,
                                 int offA, const float* PVenc, int ldenc, int offEnc, const int32_t* rank,
                                 const int32_t* E_idx32, const float* mask, const float* edge_mask,
                                 const float* W1b_packed, const float* W2_packed, const float* b2,
                                 const int32_t* node_list, int n_list, int enc_nodes, int L, int K, const float* hV_in,
                                 float cnt_const, const float* W3_packed, const float* b3, const float* ln1_w,
                                 const float* ln1_b, const float* Win_packed, const float* b_in,
                                 const float* Wout_packed, const float* b_out, const float* ln2_w, const float* ln2_b,
                                 float scale, float* hV_out, const float* Wlogit_packed, const float* blogit,
                                 const float* logit_bias, int n_letters, float temperature, uint64_t seed,
                                 const float* designable, const int64_t* S_in, int64_t* S_out, float* log_probs,
                                 const float* TS, int ld_ts, int n_layers, float* const* A_rows, const int* A_ld,
                                 const int* A_off, void* stream);

int mpnn_node_update_subset_f16(const float* hV_in, const float* gsum, int n_tiles, float cnt_const, const float* mask,
                            const float* W3_packed, const float* b3, const float* ln1_w, const float* ln1_b,
                            const float* Win_packed, const float* b_in, const float* Wout_packed, const float* b_out,
                            const float* ln2_w, const float* ln2_b, float scale, const float* Wproj_packed,
                            const float* bproj, int nproj, const int32_t* node_list, int n_list, float* hV_out,
                            float* PQ, void* stream);

/* mpnn_node_wgrad for up to 12 linears over the same n_rows in ONE launch (all linears of one node update).  Every
 * array argument is a HOST array with one entry per job (device pointers / ints); dbias[j] may be NULL. */
int mpnn_node_wgrad_batch(int n_jobs, const float* const* dY, const int* ldy, const int* out_f, const float* const* X,
                          const int* ldx, const int* in_f, float* const* dW, const int* ldw, float* const* dbias,
                          int n_rows, float* workspace, int workspace_floats, void* stream);

/* ---- output head backward and loss ---------------------------------------------------------
 * Backward of mpnn_logits_log_softmax (reference model_utils.py :527-528): d_logits = d_log_probs - exp(log_probs) *
 * rowsum(d_log_probs); d_hV [n_nodes,128] = d_logits W_out; dW [n_letters,128] += d_logits^T hV; db [n_letters] +=
 * column sums (dW / db accumulate: zero them first; db may be NULL).  W_out is the nn.Linear weight, row-major.
 * workspace: >= 4128 floats per 64 nodes (per-workgroup partials of dW | db, added in a fixed order). */
int mpnn_logits_log_softmax_bwd(const float* d_log_probs, const float* log_probs, const float* hV, const float* W_out,
                                int n_letters, int n_nodes, float* d_hV, float* dW, float* db, float* workspace,
                                int workspace_floats, void* stream);

/* loss_smoothed of the reference's training loop (train.py:63-73): t = (one_hot(S) + weight / n_letters) normalised,
 * loss [n_nodes] = -sum_a t log_probs, loss_sum = scale * sum_n loss mask (scale = 1/2000 there),
 * and, when d_log_probs is given, the gradient of loss_sum: -t mask scale.
 * workspace: >= one float per 256 nodes (per-workgroup partial sums, added in a fixed tree). */
int mpnn_loss_smoothed(const float* log_probs, const int64_t* S, const float* mask, int n_nodes, int n_letters,
                       float weight, float scale, float* loss, float* loss_sum, float* d_log_probs, float* workspace,
                       int workspace_floats, void* stream);

/* ---- batch assembly ----------------------------------------------------------------------
 * Padded model inputs for B structures out of a flat shard in device memory, i.e. what the reference's host function
 * featurize() returns (src/mpnn/data/data_utils.py:51-183) without its per-batch numpy work and host-to-device copies.
 * Shard: X_cat [T,4,3] f32 (NaN = missing atom), S_cat [T] u8 (index into "ACDEFGHIKLMNPQRSTVWYX"), chain_cat [T] i32
 * (1-based chain number in the order the chains are laid out), designed_cat [T] u8, offsets [P+1] i64 (residues of
 * structure p are [offsets[p], offsets[p+1])).  protein_ids [B] i32 picks the structures; L_max >= their longest.
 * Outputs, all [B, L_max]: X [.,4,3] f32 (NaN -> 0, :170-172), S i64 (pad 0), mask f32 (1 where all four atoms are
 * finite, :171), chain_M f32, residue_idx i64 (100 (chain-1) + position, pad -100, :57,120), chain_encoding_all i64
 * (pad 0), and lengths [B] i32. */
int mpnn_assemble_batch(const float* X_cat, const uint8_t* S_cat, const int32_t* chain_cat, const uint8_t* designed_cat,
                        const int64_t* offsets, const int32_t* protein_ids, int B, int L_max, float* X, int64_t* S,
                        float* mask, float* chain_M, int64_t* residue_idx, int64_t* chain_encoding_all,
                        int32_t* lengths, void* stream);

/* mpnn_assemble_batch with the chain order of every structure chosen PER BATCH, as featurize does (it shuffles the
 * chains of an entry every time the entry is drawn, src/mpnn/data/data_utils.py:88; chain numbers and the +100 jumps of
 * residue_idx then follow the new order, :117-120).  Row b is the concatenation of the shard's residue ranges
 * [seg_src[k], seg_src[k] + seg_len[k]) for k in [seg_ptr[b], seg_ptr[b + 1]) - the structure's chains in the order
 * wanted (device arrays; a few integers per structure, built on the host per batch).  Outputs as mpnn_assemble_batch. */
int mpnn_assemble_batch_ordered(const float* X_cat, const uint8_t* S_cat, const uint8_t* designed_cat,
                                const int32_t* seg_ptr, const int64_t* seg_src, const int32_t* seg_len, int B,
                                int L_max, float* X, int64_t* S, float* mask, float* chain_M, int64_t* residue_idx,
                                int64_t* chain_encoding_all, int32_t* lengths, void* stream);

/* mask_self [B,L,L] f32 of featurize() (data_utils.py:61,119): 0 where both residues belong to the same chain, 1
 * elsewhere (padding included). */
int mpnn_mask_self(const int64_t* chain_encoding_all, int B, int L, float* mask_self, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MPNN_HIP_H */
