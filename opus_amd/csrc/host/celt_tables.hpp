#pragma once
#include "../hip/celt_frame_types.h"
void og_fill_celt_tables(OgCeltTables *t);
