// snk_internal.h -- private interface between the translation units of libsnacc_hip.so.
// Not part of the C-ABI (include/snacc_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct snk_ctx;

// What the deflate add-on needs from a context: the resident ASCII arena.
struct SnkSeqView {
    int device;
    hipStream_t stream;
    int n;                    // resident sequences
    const uint32_t *len;      // host: length of each
    const uint32_t *boff;     // host: byte offset of each in d_bytes (zero padded behind)
    const uint8_t *d_bytes;   // device: the ASCII arena
    int dfl_kmer;             // option "deflate_kmer" (default 1): use the six-byte index in the match search
    int dfl_norestart;        // option "deflate_norestart": pair jobs do not restart from x's stored stream (testing)
    int dfl_serial;           // option "deflate_serial": per-sequence pass by one wave per sequence (testing)
};

int snk_internal_view(snk_ctx *c, SnkSeqView *v);
int snk_internal_fail(snk_ctx *c, int code, const char *msg);
// Slot holding the add-on's state; *free_fn is called when the sequences are replaced / the context dies.
void **snk_internal_dfl_slot(snk_ctx *c, void (***free_fn)(void *));
// Device-side checks of the deflate kernels since the last call (SNK_OK / SNK_E_KERNEL); defined in snk_deflate.hip.
int snk_internal_dfl_check(snk_ctx *c);
