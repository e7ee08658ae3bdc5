#pragma once
#include "fake_npp.h"
