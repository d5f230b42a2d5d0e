// Error reporting and version of libcentertrack_hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "centertrack_hip.h"

static thread_local char g_err[512] = "";

void ct_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ct_last_error(void) { return g_err; }
// 101 (round 5): ct_conv_desc grew by proj_w_packed / proj_scale / proj_shift / proj_y / proj_ldy (round 4) and the tuning
// key "dcn_offs16" became "stem_rows"; a caller built against the 100 header passes a shorter struct -- it must compare
// ct_version() with CT_ABI_VERSION of the header it was compiled against before the first descriptor call.
extern "C" int ct_version(void) { return CT_ABI_VERSION; }

// ---- tuning knobs -------------------------------------------------------------------------
enum { CT_TUNE_CONV_CFG = 0, CT_TUNE_CONV_PIPE, CT_TUNE_CONV_SMALL_TILES, CT_TUNE_SPLITK_TARGET, CT_TUNE_DCN_BN,
       CT_TUNE_CONV_KS, CT_TUNE_CONV_KS_BELOW, CT_TUNE_CONV_KS_WAVES, CT_TUNE_XCD_REMAP, CT_TUNE_HEADS_ORDER, CT_TUNE_STEM_ROWS, CT_TUNE_DCN_SLOTS, CT_TUNE_DCN_XCD, CT_TUNE_COUNT };
static int g_tune[CT_TUNE_COUNT] = {-1, 1, 256, 512, 0, -1, 512, 2048, 0, 2, 0, 1024, 1};
static const char *g_tune_names[CT_TUNE_COUNT] = {"conv_cfg", "conv_pipe", "conv_small_tiles", "splitk_target", "dcn_bn",
                                                 "conv_ks", "conv_ks_below", "conv_ks_waves", "xcd_remap", "heads_order", "stem_rows", "dcn_slots", "dcn_xcd"};

int ct_tune_get(int key) { return g_tune[key]; }

extern "C" int ct_set_tuning(const char *key, int value)
{
    for (int i = 0; i < CT_TUNE_COUNT; ++i)
        if (key && strcmp(key, g_tune_names[i]) == 0) {
            g_tune[i] = value;
            return CT_OK;
        }
    ct_set_error("ct_set_tuning: unknown key '%s'", key ? key : "(null)");
    return CT_ERR_ARG;
}

