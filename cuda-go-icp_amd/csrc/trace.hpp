// Optional roctx ranges (tracing).  With GOICP_ROCTX=1 in the environment the engine brackets its
// phases -- creation stages, a registration, every rotation batch of the BnB, every ICP run, every multi-GPU exchange --
// with roctxRangePush/Pop from librocprofiler-sdk-roctx, so `rocprofv3 --marker-trace --kernel-trace` shows the kernels
// under the phase that launched them.  The library is looked up at run time (dlopen): nothing links against it, and
// without the variable no symbol is resolved and a range costs one predictable branch.
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace goicp {

class Trace {
public:
	static Trace& get()
	{
		static Trace t;
		return t;
	}
	void push(const char* name) const { if (push_) push_(name); }
	void pop() const { if (pop_) pop_(); }
	bool on() const { return push_ != nullptr; }

private:
	Trace()
	{
		const char* e = std::getenv("GOICP_ROCTX");
		if (!e || e[0] == '0' || e[0] == '\0') return;
		void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
		if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
		if (!h) return;
		push_ = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
		pop_ = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
		if (!push_ || !pop_) push_ = nullptr, pop_ = nullptr;
	}
	int (*push_)(const char*) = nullptr;
	int (*pop_)() = nullptr;
};

struct TraceRange {
	explicit TraceRange(const char* name) { Trace::get().push(name); }
	~TraceRange() { Trace::get().pop(); }
	TraceRange(const TraceRange&) = delete;
	TraceRange& operator=(const TraceRange&) = delete;
};

}  // namespace goicp
