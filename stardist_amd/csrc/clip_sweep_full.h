// clip_sweep_full.h -- TEMPORARY: exact-join variant placeholder (same as the fast sweep).
#pragma once
#include "clip_sweep.h"
namespace sdclip {
template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ>
struct SweepFull : Sweep<MAXV, MAXIL, MAXREC> {};
}
