#include "../hip_emu.h"
