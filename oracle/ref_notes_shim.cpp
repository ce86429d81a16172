// oracle/ref_notes_shim.cpp -- TEST INFRASTRUCTURE ONLY (see vp_oracle.h).
// A C binding around the REFERENCE's own Notes class: the one translation unit of the reference that needs nothing but the standard
// library (Source/Notes.cpp includes Notes.h, which includes <vector> <string> <cmath> <algorithm> <iostream>; every other Source/*.cpp
// reaches JuceHeader.h and is unbuildable here).  oracle/Makefile compiles this file TOGETHER WITH
// /root/reference/Source/Notes.cpp where it lies (nothing of the reference is copied), into oracle/_ref/libnotes_ref.so;
// tests/test_oracle_ref_notes.py checks the restatement (vpo_notes_build / vpo_notes_closest, vp_oracle.c) against it bit for bit.
#include "Notes.h"

extern "C" {
void *refnotes_new() { return new Notes(); }
void refnotes_free(void *p) { delete static_cast<Notes *>(p); }
// Notes::prepare (Notes.cpp:24-37)
void refnotes_prepare(void *p, int key, double fMin, double fMax) { static_cast<Notes *>(p)->prepare(static_cast<Notes::key>(key), fMin, fMax); }
// Notes::getClosestFreq (Notes.cpp:79-110): rebuilds the table when the key differs from the current one
double refnotes_closest(void *p, double pitch, int key) { return static_cast<Notes *>(p)->getClosestFreq(pitch, static_cast<Notes::key>(key)); }
}
