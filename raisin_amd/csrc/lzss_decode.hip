#include "codecs.h"
namespace rsn {
int lzss_decode_dev(Ctx &c, hipStream_t, const uint8_t *, size_t, uint8_t *, size_t, size_t *) { return c.fail(RSN_ERR_LIMIT, "lzss decode: not built yet"); }
}
