// snk_device.hip.h -- umbrella include of the gfx950 device code (see snk_common.hip.h).
#pragma once
#include "snk_common.hip.h"
#include "snk_fast.hip.h"
#include "snk_legacy.hip.h"
#include "snk_bytes.hip.h"
#include "snk_ingest.hip.h"
#include "snk_emit.hip.h"
