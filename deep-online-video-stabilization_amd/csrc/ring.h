// Launchers of the on-device history ring (ring.hip).
#pragma once
#include "common.h"

struct RingLags { int lag[8]; int n; };

int launch_ring_init(float* frames, float* masks, const float* first, int S, int depth, long hw, hipStream_t st);
int launch_stack_assemble(const float* frames, const float* masks, const float* cur, int S, int depth, const int* head,
                          const RingLags& lags, long hw, int Cp, float* out, hipStream_t st);
int launch_stack_assemble_bordered(const float* frames, const float* masks, const float* cur, int S, int depth, const int* head,
                                   const RingLags& lags, int H, int W, int border, float* out, hipStream_t st);
int launch_embed_border(const float* x, int N, int H, int W, int C, int border, float* out, hipStream_t st);
int launch_ring_push(float* frames, float* masks, int S, int depth, const int* head, const float* img, const float* black,
                     long hw, float* frame_out, hipStream_t st);
int launch_ring_advance(int* head, int depth, hipStream_t st);
