import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], d["config"]["step_mix"]["critic_ms"], d["config"]["step_mix"]["generator_ms"], d["config"]["step_mix"].get("critic_ms_each"), d["config"]["step_mix"].get("device_allocations_in_window"))
