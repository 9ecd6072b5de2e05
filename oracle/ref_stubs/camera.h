// ORACLE — TEST INFRASTRUCTURE ONLY.  Empty stand-in: src/line_processor.cc:1-180 uses nothing of camera.h.
#pragma once
