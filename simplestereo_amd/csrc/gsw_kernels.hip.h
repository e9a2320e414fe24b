// K3/K4: GSW kernels (filled in after the ASW path is verified on hardware).
#pragma once
#include "common.hip.h"
