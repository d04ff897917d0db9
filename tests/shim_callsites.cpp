// Compile-only check of the C++ shim (tests/test_host_boundary.py).  Each function below has the SHAPE of one of the
// reference's call sites for this path -- the types on the caller's side are the reference's (glm with
// -DSHIM_WITH_GLM), the callee is include/goicp_mi355.hpp.  Nothing here is executed.
#ifdef SHIM_WITH_GLM
#include <glm/glm.hpp>
#endif
#include <mutex>
#include <thread>
#include <vector>

#include "goicp_mi355.hpp"

using namespace goicp_mi355;
#ifdef SHIM_WITH_GLM
using mat3 = glm::mat3;
using vec3 = glm::vec3;
#else
using mat3 = Mat3;
using vec3 = Vec3;
#endif

// the viewer's globals (src/main.cpp:20-31)
static std::mutex mtx;
static std::vector<vec3> dataBuffer, modelBuffer;
static icp::FastGoICP* fgoicp = nullptr;
static mat3 prev_optR_fgoicp;
static vec3 prev_optT_fgoicp;
static float mse_threshold = 1e-3f;

// src/main.cpp:33-35: Config + load_cloud
void load(const char* path)
{
	Config config(path);
	load_cloud(config.io.source, config.subsample, config.resize, dataBuffer);
	load_cloud(config.io.target, config.subsample, config.resize, modelBuffer);
	mse_threshold = config.mse_threshold;
}
// src/main.cpp:79,94: buffers + engine construction
void init()
{
	PointCloud::initBuffers(dataBuffer, modelBuffer);
	fgoicp = new icp::FastGoICP(modelBuffer, dataBuffer, mse_threshold, mtx);
}
// src/main.cpp:150-151: the worker thread
void start_worker()
{
	std::thread fgoicp_thread(&icp::FastGoICP::run, fgoicp);
	fgoicp_thread.detach();
}
// src/main.cpp:99-141: the step dispatch
void run_step(int mode)
{
	int kdtree = 0, tree = 0;
	int* dev_fkdt = nullptr;
	switch (mode) {
	case 0: ICP::CPUStep(dataBuffer, modelBuffer); break;
	case 1: ICP::naiveGPUStep(); break;
	case 2: ICP::kdTreeGPUStep(kdtree, tree, dev_fkdt); break;
	default: ICP::goicpGPUStep(fgoicp, prev_optR_fgoicp, prev_optT_fgoicp, mtx); break;
	}
}
// src/goicp_kernel.cu:161-177: the viewer's read of the result API, the caller's side in the reference's types
bool poll_like_the_viewer(const icp::FastGoICP* f, mat3& prev_optR, vec3& prev_optT, mat3& curR, vec3& curT, float& currentError)
{
	bool updated, finished;
	{
		std::lock_guard<std::mutex> lock(mtx);
		finished = f->finished;
		updated = (prev_optR != f->optR || prev_optT != f->optT);
		currentError = f->get_best_error();
		prev_optR = f->optR;
		prev_optT = f->optT;
		curR = f->curR;
		curT = f->curT;
	}
	return updated || finished;
}
// src/fgoicp/registration.hpp:96-97, icp3d.hpp:30-35, fgoicp.cpp:11-12,140
float operators(icp::Registration& reg)
{
	icp::RotNode rnode(0.1f, 0.2f, 0.3f, 0.05f, 0.f, 1e10f);
	std::vector<icp::TransNode> tnodes;
	tnodes.emplace_back(0.f, 0.f, 0.f, 0.25f, 0.f, 1e10f);
	icp::StreamPool stream_pool(32);
	auto [lb, ub] = reg.compute_sse_error(rnode, tnodes, true, stream_pool);
	const float sse0 = reg.compute_sse_error(mat3(1.0f), vec3(0.0f));
	icp::IterativeClosestPoint3D icp3d(reg, modelBuffer, dataBuffer, 1000, 1e-7f, mat3(1.0f), vec3(0.0f));
	mat3 curR;
	vec3 curT;
	auto [sse, R, t] = icp3d.run(curR, curT);
	(void)R; (void)t;
	return lb[0] + ub[0] + sse0 + sse;
}
