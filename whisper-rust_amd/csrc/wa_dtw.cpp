// wa_dtw.cpp - DTW token-level timestamps.  Filled in after the greedy
// path (rows a1-a13) is parity-green; until then the request is accepted and reported as unsupported.
#include "wa_internal.h"

void wa_dtw_timestamps(whisper_context *, whisper_state *, const whisper_full_params &, int, size_t, int, int, int) {
    WA_WARN("%s: DTW token timestamps are not implemented yet on this backend; t_dtw stays -1\n", __func__);
}
