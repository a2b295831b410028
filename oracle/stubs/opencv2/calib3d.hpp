// TEST INFRASTRUCTURE: declaration-free stand-in so that the reference's model.cpp (which never
// names a cv:: type itself) can be compiled in place without OpenCV (oracle/Makefile, target ref).
#pragma once
