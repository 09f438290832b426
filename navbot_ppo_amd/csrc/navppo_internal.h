// navppo_internal.h -- shared by the translation units of libnavsim.so that implement include/navppo.h (not installed)
#pragma once

// stores the message navppo_last_error() returns (thread-local, defined in ppo_mlp64.hip)
void navppo_set_error(const char* msg);
