/* oracle/ref_build/run_accel_shim.c -- TEST INFRASTRUCTURE ONLY.
 * Exports the reference's run_accel (src/nfa/accel.c:35-146, compiled in place from the reference tree) on a raw
 * 80-byte union AccelAux image: the offset, relative to buf, of the point the scheme says nothing matches before,
 * scanning [buf + start, buf + len). What hsgpu_run_accel_dev is checked against. */
#include <string.h>

#include "ue2common.h"
#include "nfa/accel.h"

size_t hsref_run_accel(const unsigned char img[80], const unsigned char *buf, size_t len, size_t start) {
    union AccelAux aux;
    memcpy(&aux, img, sizeof(aux) < 80 ? sizeof(aux) : 80);
    return (size_t)(run_accel(&aux, buf + start, buf + len) - buf);
}
