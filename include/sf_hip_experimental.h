/* Entry points of libsf_experimental.so -- experiments that are correct but slower than the product
 * path, built on demand (`python -m speaker_follower_amd.build --experimental`) and NOT part of
 * libsf_hip.so or its ABI version.  Types: sf_hip.h. */
#pragma once
#include "sf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The S decode steps of an INFERENCE rollout (no dropout, no backward to follow) as ONE persistent launch
 * (csrc/sf_mega.hip): same inputs and outputs as sf_follower_episode_fwd -- logits, actions, scores, CE
 * terms, liveness, `ended`, the h1 / c1 tapes -- without the per-step tapes of the backward.  Needs
 * w->fold (sf_decoder_fold_build), index-form panoramas / candidates (no dense tensors, no is_valid),
 * B <= 128, H = 512, F = 2176, V = 36, L <= 80, A <= 16; anything else returns SF_ERR_UNSUPPORTED and
 * the caller uses sf_follower_episode_fwd.  debug_tapes != 0 (tests): t_text, cat2[:, :H], h_tilde,
 * q and xin of every step are also copied into e->tape. */
int sf_follower_decode_persistent(const sf_decoder_w* w, const sf_follower_episode* e, int debug_tapes,
                                  void* ws, size_t ws_bytes, sf_stream stream);

#ifdef __cplusplus
}
#endif
