#pragma once
#include "npp_over_vpf.h"
