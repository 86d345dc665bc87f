// ONE place for every environment switch of libvitx ("42 getenv sites ... no single place that says which are safe").
// Every read of a VITX_* variable in csrc/ goes through vitx_env(); the table in env.hip classifies each name:
//   TUNING  same results (bit-identical, or the same value up to the order of a fixed-order fp32 sum): another schedule, stream layout or launch shape
//   PATH    selects another VALIDATED code path (its own tests against the same oracle gates; results agree within those gates, not bit for bit)
//   DIAG    timing experiment / diagnostic that MAY CORRUPT RESULTS (a K loop without its DMA wait, a kernel without one of its stages, zero operands):
//           honoured only by a diagnostic build (-DVITX_DIAG, `python vit-tensorflow_amd/build.py --diag` -> lib/libvitx_diag.so); the release
//           library ignores it and says so once on stderr.
// A name that is not in the table is a programming error (vitx_env aborts): a new switch cannot be added without classifying it.
#pragma once

enum VitxEnvClass { VITX_ENV_TUNING = 0, VITX_ENV_PATH = 1, VITX_ENV_DIAG = 2 };
struct VitxEnvSwitch { const char* name; int cls; const char* doc; };

const VitxEnvSwitch* vitx_env_table(int* n);
// getenv(name) through the registry; nullptr when unset -- and, in a release build, for every DIAG switch
const char* vitx_env(const char* name);
inline bool vitx_env_flag(const char* name) { const char* v = vitx_env(name); return v && v[0] && v[0] != '0'; }
// 1 in a -DVITX_DIAG build
int vitx_env_diag_build();
