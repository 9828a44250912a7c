// Error plumbing + version of the C ABI (include/mpnn_hip.h).  No exceptions cross the ABI:
// launchers return 0 or a non-zero code and leave a message here.
#include <string.h>

static thread_local char g_err[256] = "";

extern "C" void mpnn_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* mpnn_last_error(void) { return g_err; }

extern "C" int mpnn_version(void) { return 1; }

// Workspace query (include/mpnn_hip.h).  16512 floats = one 128 x 128 partial + 128 bias sums.
extern "C" long long mpnn_workspace_bytes(const char* entry, int n_rows, int n_blocks) {
  const long long part = 16512LL * (long long)sizeof(float);
  if (!entry || n_rows < 0 || n_blocks < 0) return -1;
  if (!strcmp(entry, "mpnn_edge_wgrad")) {  // one partial per workgroup, at most 3 per CU of the largest part (256 CUs)
    const long long wg = n_rows < 768 ? (n_rows > 0 ? n_rows : 1) : 768;
    return wg * part;
  }
  if (!strcmp(entry, "mpnn_message_bwd_fused")) return (64 + 256 * 2 * 16512LL) * (long long)sizeof(float);
  if (!strcmp(entry, "mpnn_enc_edge_update_bwd_fused")) return (64 + 256 * (2 * 16512LL + 256)) * (long long)sizeof(float);
  if (!strcmp(entry, "mpnn_node_wgrad") || !strcmp(entry, "mpnn_node_wgrad_batch")) {
    // one partial per 128 x 128 output block and row chunk; chunks of >= 32 rows, at most 256 of them
    long long chunks = (n_rows + 31) / 32;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    return chunks * (n_blocks > 0 ? n_blocks : 1) * part;
  }
  // per-workgroup partial sums that a second small launch adds in a fixed order (no floating-point atomics)
  const long long f = (long long)sizeof(float);
  if (!strcmp(entry, "mpnn_enc_edge_update_bwd") || !strcmp(entry, "mpnn_edge_features_bwd"))
    return 1024 * 256 * f;  // 256 floats per workgroup, one workgroup per CU at most
  if (!strcmp(entry, "mpnn_node_update_bwd")) return ((n_rows + 63) / 64) * 640LL * f;
  if (!strcmp(entry, "mpnn_logits_log_softmax_bwd")) return ((n_rows + 63) / 64) * 4128LL * f;
  if (!strcmp(entry, "mpnn_loss_smoothed")) return ((n_rows + 255) / 256 + 1) * f;
  if (!strcmp(entry, "mpnn_gather_rows")) return (n_blocks > 0 ? n_blocks : 1) * 256LL * f;  // n_blocks = max_chunks
  if (!strcmp(entry, "mpnn_grad_sumsq"))  // n_rows = the largest tensor's elements, n_blocks = tensors
    return ((n_rows + 2047) / 2048) * (long long)(n_blocks > 0 ? n_blocks : 1) * f;
  if (!strncmp(entry, "mpnn_", 5)) return 0;  // every other entry point works in the caller's inputs and outputs only
  return -1;
}

// 1 when the library was built with a timing-only ablation switch (mpnn_ablation.h): its outputs are wrong by design
#include "mpnn_ablation.h"
extern "C" int mpnn_ablation_build(void) { return MPNN_ABLATION_BUILD; }
