#pragma once
#include "cuda.h"
