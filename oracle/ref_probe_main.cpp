// TEST INFRASTRUCTURE ONLY (oracle/). Never linked into, imported by, or executed from the product path.
//
// vxprobe: a state-dumping driver around the UNMODIFIED reference Voxelyze sources (compiled where
// they lie under /root/reference by oracle/Makefile into oracle/_ref/). It runs exactly the loop of the
// reference headless main (evosoro/_voxcad/voxelyzeMain/main.cpp:89-111) and additionally writes a
// binary trace of the full voxel state so that the C restatement (oracle/vx_oracle.c) and the HIP engine
// can be compared step by step against the real reference, not only through the 6-digit result XML.
//
// usage: vxprobe -f in.vxa -o trace.bin [-every K] [-max N] [-noresult]
// trace format (little endian):
//   header : int32 magic 0x56585452, int32 nvox, int32 nbond, double dt_opt (OptimalDt), double dtfrac
//   record : int32 step (steps completed), int32 ncol, double curtime, double dt, double cm[3],
//            then nvox * 14 doubles: pos3, quat(w,x,y,z), scale, vel3, angvel3
//   the last record is always the final state; int32 step = -1 terminates the file followed by
//   double inicm[3], double curcm[3], int32 total_steps.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <iostream>
#include "VX_Object.h"
#include "VX_Environment.h"
#include "VX_Sim.h"
#include "VX_SimGA.h"
#include "VX_MeshUtil.h"

static void dump(FILE* f, CVX_SimGA& S)
{
	int step = S.CurStepCount, ncol = S.NumColBond();
	fwrite(&step, 4, 1, f); fwrite(&ncol, 4, 1, f);
	double t = S.CurTime, dt = S.dt;
	fwrite(&t, 8, 1, f); fwrite(&dt, 8, 1, f);
	Vec3D<> cm = S.NumVox() ? S.GetCM() : Vec3D<>(0,0,0);
	double c[3] = {cm.x, cm.y, cm.z}; fwrite(c, 8, 3, f);
	for (int i = 0; i < S.NumVox(); i++) {
		CVXS_Voxel& v = S.VoxArray[i];
		Vec3D<double> p = v.GetCurPosHighAccuracy(); CQuat<double> q = v.GetCurAngleHighAccuracy();
		Vec3D<> vel = v.GetCurVel(), w = v.GetCurAngVel();
		double r[14] = {p.x, p.y, p.z, q.w, q.x, q.y, q.z, v.GetCurScale(), vel.x, vel.y, vel.z, w.x, w.y, w.z};
		fwrite(r, 8, 14, f);
	}
}

int main(int argc, char* argv[])
{
	const char* in = 0; const char* out = 0; long every = 1, maxsteps = -1; bool noresult = false;
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "-f") && i + 1 < argc) in = argv[++i];
		else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
		else if (!strcmp(argv[i], "-every") && i + 1 < argc) every = atol(argv[++i]);
		else if (!strcmp(argv[i], "-max") && i + 1 < argc) maxsteps = atol(argv[++i]);
		else if (!strcmp(argv[i], "-noresult")) noresult = true;
	}
	if (!in || !out) { fprintf(stderr, "usage: vxprobe -f in.vxa -o trace.bin [-every K] [-max N] [-noresult]\n"); return 2; }

	CVX_Object Object; CVX_Environment Environment; CVX_SimGA Simulator; CVX_MeshUtil DeformableMesh;
	Simulator.pEnv = &Environment; Environment.pObj = &Object; Simulator.setInternalMesh(&DeformableMesh);
	if (!Simulator.LoadVXAFile(in)) { fprintf(stderr, "load failed\n"); return 3; }
	std::string msg;
	Simulator.Import(&Environment, 0, &msg);
	if (Simulator.NumVox() == 0) { fprintf(stderr, "no voxels\n"); return 4; }
	vfloat Time = 0.0;
	Simulator.pEnv->UpdateCurTemp(Time);
	DeformableMesh.initializeDeformableMesh(&Simulator);

	FILE* f = fopen(out, "wb"); if (!f) return 5;
	int magic = 0x56585452, nv = Simulator.NumVox(), nb = Simulator.NumBond();
	fwrite(&magic, 4, 1, f); fwrite(&nv, 4, 1, f); fwrite(&nb, 4, 1, f);
	double od = Simulator.OptimalDt, df = Simulator.DtFrac; fwrite(&od, 8, 1, f); fwrite(&df, 8, 1, f);
	dump(f, Simulator);

	long Step = 0;
	while (!Simulator.StopConditionMet()) {
		if (maxsteps >= 0 && Step >= maxsteps) break;
		bool ok = Simulator.TimeStep(&msg);
		Step++; Time += Simulator.dt; Simulator.pEnv->UpdateCurTemp(Time);
		if (!ok) { fprintf(stderr, "diverged at step %ld\n", Step); break; } // the reference main would spin forever here
		if (Step % every == 0) dump(f, Simulator);
	}
	if (Step % every != 0) dump(f, Simulator);
	int endm = -1; fwrite(&endm, 4, 1, f);
	double a[3] = {Simulator.IniCM.x, Simulator.IniCM.y, Simulator.IniCM.z}; fwrite(a, 8, 3, f);
	double b[3] = {Simulator.SS.CurCM.x, Simulator.SS.CurCM.y, Simulator.SS.CurCM.z}; fwrite(b, 8, 3, f);
	int ts = Simulator.CurStepCount; fwrite(&ts, 4, 1, f);
	fclose(f);
	if (!noresult && Simulator.FitnessFileName != "") Simulator.SaveResultFile(Simulator.FitnessFileName);
	return 0;
}
