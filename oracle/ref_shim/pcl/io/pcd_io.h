// ORACLE — TEST INFRASTRUCTURE ONLY.  Empty stand-in: relocator.cpp includes this header and uses nothing of it.
#pragma once
