/* oracle/ref_build/accel_block_shim.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's pre-skip, do_accel_block (src/hwlm/hwlm.c:80-99), is a static function: this file includes the
 * reference's hwlm.c where it lies (nothing is copied) under other names for its two public entry points, and exports
 * a wrapper, so that the pre-skip the GPU library computes (hsgpu_hwlm_forward_skip_dev) can be checked against the
 * reference's own code rather than against a restatement of it.
 * `img`: the decoded AccelAux of hsref_forward_accel (ref_driver.cpp): [0] type [1] offset [2] c / c1 [3] c2
 * [16..32) lo / mask1 [48..64) hi / mask2. */
#define hwlmExec hsref_shim_unused_hwlmExec
#define hwlmExecStreaming hsref_shim_unused_hwlmExecStreaming
#include "hwlm/hwlm.c"

size_t hsref_do_accel_block(const unsigned char img[80], const unsigned char *buf, size_t len, size_t start) {
    union AccelAux aux;
    memset(&aux, 0, sizeof(aux));
    aux.accel_type = img[0];
    aux.generic.offset = img[1];
    switch (img[0]) {
    case ACCEL_VERM:
    case ACCEL_VERM_NOCASE:
        aux.verm.c = img[2];
        break;
    case ACCEL_DVERM:
    case ACCEL_DVERM_NOCASE:
        aux.dverm.c1 = img[2];
        aux.dverm.c2 = img[3];
        break;
    case ACCEL_SHUFTI:
        memcpy(&aux.shufti.lo, img + 16, 16);
        memcpy(&aux.shufti.hi, img + 48, 16);
        break;
    case ACCEL_TRUFFLE:
        memcpy(&aux.truffle.mask1, img + 16, 16);
        memcpy(&aux.truffle.mask2, img + 48, 16);
        break;
    default:
        break;
    }
    do_accel_block(&aux, buf, len, &start);
    return start;
}
