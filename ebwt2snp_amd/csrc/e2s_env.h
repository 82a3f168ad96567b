// e2s_env.h -- the library's environment switches, in two tiers.
//   supported (plain getenv; documented in DESIGN.md section 4, each exercised by the test suite or the tools):
//     E2S_TIMING, E2S_CLI_TIMING, E2S_CLI_JSON, E2S_DEBUG_SYNC, E2S_CHUNK_ENTRIES, E2S_TAIL_THREADS, E2S_UPLOAD_THREADS,
//     E2S_MGPU_FULL_COLUMNS, E2S_MGPU_GATHER_READS, E2S_MGPU_SHARED_DEVICE, E2S_MGPU_PREAD (the driver's shard uploads with pread into pinned buffers instead of from mappings), E2S_MGPU_STREAM_READS (a lone rank streams the FASTA), E2S_MGPU_STAGING=0 (the driver's copies out of mappings by the runtime's staging),
//     E2S_READS_MAPPED (e2s_upload_reads_fasta over a mapping instead of the streamed pass), E2S_FASTA_TWO_PASS (e2s_fasta_index_open: the two-pass index always)
//   development (e2s_env_dev: read ONLY when E2S_DEV=1 -- tuning aids and the test suite's ways of forcing rare paths; a production
//   process that happens to have one of them set is not affected): everything else.
#pragma once
#include <stdlib.h>
static inline bool e2s_dev_mode() { static const bool on = getenv("E2S_DEV") != nullptr && atoi(getenv("E2S_DEV")) != 0; return on; }
static inline const char* e2s_env_dev(const char* name) { return e2s_dev_mode() ? getenv(name) : nullptr; }
