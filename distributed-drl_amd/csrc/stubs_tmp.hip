// TEMPORARY during bring-up: symbols not implemented yet report DDRL_ERR_UNSUPPORTED.
#include "ddrl_common.h"
#define STUB(name, ...) int name(__VA_ARGS__) { ddrl::set_error(#name " not implemented yet"); return DDRL_ERR_UNSUPPORTED; }
extern "C" {
STUB(ddrl_env_create, ddrl_env_t **, int, int64_t, uint32_t, int32_t)
STUB(ddrl_env_destroy, ddrl_env_t *)
STUB(ddrl_env_reset, ddrl_env_t *, const uint8_t *, float *, void *)
STUB(ddrl_env_step, ddrl_env_t *, const float *, float *, float *, float *, float *, uint8_t *, void *)
STUB(ddrl_env_stats, ddrl_env_t *, int64_t *, double *, int64_t *, void *)
STUB(ddrl_env_get_state, ddrl_env_t *, float *, void *)
STUB(ddrl_env_set_state, ddrl_env_t *, const float *, void *)
}
