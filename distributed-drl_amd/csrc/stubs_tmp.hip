// TEMPORARY during bring-up: symbols not implemented yet report DDRL_ERR_UNSUPPORTED.
#include "ddrl_common.h"
#define STUB(name, ...) int name(__VA_ARGS__) { ddrl::set_error(#name " not implemented yet"); return DDRL_ERR_UNSUPPORTED; }
extern "C" {
STUB(ddrl_sac1_param_counts, const ddrl_sac1_config_t *, int64_t *, int64_t *)
STUB(ddrl_sac1_create, ddrl_sac1_t **, int, const ddrl_sac1_config_t *)
STUB(ddrl_sac1_destroy, ddrl_sac1_t *)
STUB(ddrl_sac1_set_weights, ddrl_sac1_t *, const float *, void *)
STUB(ddrl_sac1_get_weights, ddrl_sac1_t *, float *, void *)
STUB(ddrl_sac1_state, ddrl_sac1_t *, float **, float **, float **, float **, int64_t *, int64_t *)
STUB(ddrl_sac1_step, ddrl_sac1_t *, const float *, const float *, const float *, const float *, const float *, const float *, const float *, const float *, float *, float *, float *, float *, void *)
STUB(ddrl_sac1_grads, ddrl_sac1_t *, float **, int64_t *)
STUB(ddrl_sac1_compute_grads, ddrl_sac1_t *, const float *, const float *, const float *, const float *, const float *, const float *, const float *, const float *, float *, float *, float *, float *, void *)
STUB(ddrl_sac1_apply_grads, ddrl_sac1_t *, void *)
STUB(ddrl_actor_create, ddrl_actor_t **, int, const ddrl_sac1_config_t *, int64_t)
STUB(ddrl_actor_destroy, ddrl_actor_t *)
STUB(ddrl_actor_set_weights, ddrl_actor_t *, const float *, void *)
STUB(ddrl_actor_params, ddrl_actor_t *, float **, int64_t *)
STUB(ddrl_actor_act, ddrl_actor_t *, const float *, const float *, int64_t, int, float *, void *)
STUB(ddrl_env_create, ddrl_env_t **, int, int64_t, uint32_t, int32_t)
STUB(ddrl_env_destroy, ddrl_env_t *)
STUB(ddrl_env_reset, ddrl_env_t *, const uint8_t *, float *, void *)
STUB(ddrl_env_step, ddrl_env_t *, const float *, float *, float *, float *, float *, uint8_t *, void *)
STUB(ddrl_env_stats, ddrl_env_t *, int64_t *, double *, int64_t *, void *)
STUB(ddrl_env_get_state, ddrl_env_t *, float *, void *)
STUB(ddrl_env_set_state, ddrl_env_t *, const float *, void *)
}
