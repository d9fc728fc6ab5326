// SILK parse half (placeholder until the SILK kernels land): reports OPUSGPU_UNSUPPORTED.
#include "parse.hpp"
#include "../../../include/opusgpu.h"
namespace og {
void SilkChanHost::flush() { *this = SilkChanHost{}; }
void SilkHostState::flush() { mid.flush(); side.flush(); stereo_reset = true; }
int silk_parse(SilkHostState &, RangeDec &, int, int, int, std::vector<OgSilkFrame> &) { return OPUSGPU_UNSUPPORTED; }
}  // namespace og
