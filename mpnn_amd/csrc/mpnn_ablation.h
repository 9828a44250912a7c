// The timing-only ablation switches of the kernels, in one place.  Each -DMPNN_ABL_* removes ONE ingredient of a
// kernel so that what it costs can be read off an A/B run (tools/build_variant.sh, tools/ab_bench.sh; results in
// profiles/r0*_ablation.txt, r02_ab_abl.txt, r02_sampler_levels.txt, r02_train_experiments.txt).  Their outputs are
// WRONG by construction: a library built with any of them reports mpnn_ablation_build() = 1 and the Python binding
// refuses to load it unless MPNN_ALLOW_ABLATION=1 is set.  `make check-switches` compiles every switch once.
//
//   common (mpnn_common.h, mpnn_edge_common.h)
//     MPNN_ABL_NOGELU         GELU replaced by the identity
//     MPNN_ABL_NOREDUCE       column sums (DPP reductions) skipped
//     MPNN_ABL_NOMFMA         operands stay live, the matrix pipe is never used
//     MPNN_ABL_WSTREAM_SAME   every streamed weight fragment read hits the same cache lines
//     MPNN_ABL_WSTREAM_NONE   no weight fragment is streamed at all
//     MPNN_ABL_NOROW          gathered neighbour rows not loaded
//   message / edge update (mpnn_edge.hip)
//     MPNN_ABL_NOXLOAD        hE k-steps not loaded
//     MPNN_ABL_XSAME          every hE read hits L2 (64 residues' blocks)
//     MPNN_ABL_ROWSAME        all 16 columns gather the residue's own row
//     MPNN_ABL_W3LDS          edge update: third matrix entirely from LDS (no L2 stream)
//     MPNN_ABL_NORES          edge update: no residual re-read
//     MPNN_ABL_NOLN           edge update: no LayerNorm
//     MPNN_ABL_NOSTORE        edge update: no hE store
//   sampler level kernels (mpnn_edge.hip, mpnn_node.hip, mpnn_sample.h)
//     MPNN_ABL_LEVEL_NOFILL   message weights not staged to LDS
//     MPNN_ABL_LEVEL_NOMSG    no message phase
//     MPNN_ABL_LEVEL_NONODE   no node phase
//     MPNN_ABL_SAMPLE_NOGEMM  head: no logits GEMM
//     MPNN_ABL_SAMPLE_NOLOOP  head: no inverse-CDF loop
//     MPNN_ABL_SAMPLE_NOTAIL  head: sequence terms not folded into the gathered rows
//   node update (mpnn_node.hip)
//     MPNN_ABL_NODMA          matrices not staged by LDS-DMA
//   weight gradient (mpnn_edge_bwd.hip)
//     MPNN_ABL_WGRAD_STREAM   operand streams alone (no LDS transposes, no MFMAs)
//   fused backward kernels (mpnn_edge_bwd.hip: k_message_bwd_fused, k_edge_update_bwd_fused)
//     MPNN_ABL_FUSED_SAMELOAD every operand tile read hits the workgroup's first residue (L2): no HBM read latency
//     MPNN_ABL_FUSED_NOBARRIER no workgroup barriers (races: what the lock-step costs)
//
// (The measured-and-rejected kernel VARIANTS - tile-wise kernels at K <= 48, burst prefetches, LDS read-ahead rings,
// staggered half-workgroups, alternative load placements, the whole-level sampler kernel, the fp32-MFMA node weight
// gradient, ... - are not switches any more: they were deleted in round 3, see profiles/README.md.)
#pragma once

#if defined(MPNN_ABL_NOGELU) || defined(MPNN_ABL_NOREDUCE) || defined(MPNN_ABL_NOMFMA) ||                        \
    defined(MPNN_ABL_WSTREAM_SAME) || defined(MPNN_ABL_WSTREAM_NONE) || defined(MPNN_ABL_NOROW) ||              \
    defined(MPNN_ABL_NOXLOAD) || defined(MPNN_ABL_XSAME) || defined(MPNN_ABL_ROWSAME) || defined(MPNN_ABL_W3LDS) || \
    defined(MPNN_ABL_NORES) || defined(MPNN_ABL_NOLN) || defined(MPNN_ABL_NOSTORE) ||                           \
    defined(MPNN_ABL_LEVEL_NOFILL) || defined(MPNN_ABL_LEVEL_NOMSG) || defined(MPNN_ABL_LEVEL_NONODE) ||        \
    defined(MPNN_ABL_SAMPLE_NOGEMM) || defined(MPNN_ABL_SAMPLE_NOLOOP) || defined(MPNN_ABL_SAMPLE_NOTAIL) ||    \
    defined(MPNN_ABL_NODMA) || defined(MPNN_ABL_WGRAD_STREAM) || defined(MPNN_ABL_FUSED_SAMELOAD) ||             \
    defined(MPNN_ABL_FUSED_NOBARRIER)
#define MPNN_ABLATION_BUILD 1
#else
#define MPNN_ABLATION_BUILD 0
#endif
