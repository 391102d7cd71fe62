// hep_knobs.cpp - the only translation unit of the plan builder that reads the environment (hep_knobs.h).
#include "hep_knobs.h"

#include <stdlib.h>
#include <string.h>

namespace hep {

static void env_int(const char* name, int* v) { if (const char* e = getenv(name)) *v = atoi(e); }

Knobs read_knobs() {
  Knobs k;
  env_int("HEP_LANES", &k.lanes);
  env_int("HEP_CHAIN_STREAM", &k.chain_stream);
  env_int("HEP_TOWER_COOP", &k.tower_coop);
  if (const char* e = getenv("HEP_MBF_MP")) k.mbf_mp = !strcmp(e, "force") ? 3 : (atoi(e) == 2 ? 2 : (atoi(e) != 0 ? 1 : 0));
  env_int("HEP_XBF_GENERIC", &k.xbf_generic);
  env_int("HEP_STEM_MFMA", &k.stem_mfma);
  if (const char* e = getenv("HEP_SE_MAXMB")) k.se_maxmb = atof(e);
  env_int("HEP_PW_FRAG", &k.pw_frag);
  env_int("HEP_SEP_WLDS", &k.sep_wlds);
  env_int("HEP_PLAN_DEBUG", &k.plan_debug);
#ifdef HEP_ALT
  env_int("HEP_PW_NT2", &k.pw_nt2);
  env_int("HEP_PW_MT2", &k.pw_mt2);
  env_int("HEP_PW_NT3", &k.pw_nt3);
  env_int("HEP_PW_WIDE", &k.pw_wide);
  env_int("HEP_PW_W8", &k.pw_w8);
  env_int("HEP_PW_W8_MINK", &k.pw_w8_mink);
  env_int("HEP_SE_TAIL", &k.se_tail);
  env_int("HEP_XBF", &k.xbf);
  env_int("HEP_XBF_MINH", &k.xbf_minh);
  env_int("HEP_XBF_TPW", &k.xbf_tpw);
  if (const char* e = getenv("HEP_MBF")) k.mbf = !strcmp(e, "all") ? 1 : (!strcmp(e, "none") ? 0 : -1);
  env_int("HEP_MBF_MAXH", &k.mbf_maxh);
  if (const char* e = getenv("HEP_MBF_TS")) k.mbf_ts8 = atoi(e) == 8;
  env_int("HEP_MBF_TS16_MAXH", &k.mbf_ts16_maxh);
  env_int("HEP_MBF_CC", &k.mbf_cc);
  env_int("HEP_MBF_MP_RES", &k.mbf_mp_res);
  env_int("HEP_DWLDS", &k.dwlds);
  env_int("HEP_LATE", &k.late);
  env_int("HEP_LATE_G", &k.late_g);
  env_int("HEP_LATE_XCD", &k.late_xcd);
  env_int("HEP_HEADS_FUSED", &k.heads_fused);
  env_int("HEP_SBF", &k.sbf);
  env_int("HEP_TOWER", &k.tower);
  env_int("HEP_CHAIN", &k.chain);
  env_int("HEP_CHAIN_F32", &k.chain_f32);
  env_int("HEP_CHAIN_WGLOBAL", &k.chain_wglobal);
  env_int("HEP_PWG", &k.pwg);
  env_int("HEP_SEP_TS4_MAXHW", &k.sep_ts4_maxhw);
#endif
  return k;
}

}  // namespace hep
