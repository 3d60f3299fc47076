#include "common.hpp"
mrl_sim *mrl::create_hanabi(const mrl_hanabi_config *, int, uint32_t)
{
    set_error("hanabi: not built yet");
    throw HipError{MRL_ERR_INVALID};
}
