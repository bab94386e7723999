// ws_march_nd4.hip -- the marching kernel with 4 disparities per thread (see ws_march_kernel.h for why it is a
// translation unit of its own).  Part of the gfx950 kernels of the WindowSearch hot path; overview in ws_march.hip.
#include "ws_march_kernel.h"

namespace wsamd {

static const MarchEntry kMarchNarrow[] = {WS_MARCH_TABLE(kNDNarrow, ",nd4")};

const MarchEntry *march_table_narrow(int *count)
{
    *count = (int)(sizeof kMarchNarrow / sizeof kMarchNarrow[0]);
    return kMarchNarrow;
}

} // namespace wsamd
