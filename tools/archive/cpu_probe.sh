# host CPU of the plain C++ host program (examples/native_units, no Python / torch in the process) under a few runtime settings
cd $GRAFT_REPO_ROOT
python tools/export_artifacts.py /tmp/art 20 > /dev/null 2>&1
make -C examples > /dev/null
for envs in "GL355_NONE=1"; do
  echo "== env: $envs"
  ( export $envs; TIMEFORMAT="wall %R s, user %U s, sys %S s"; time ./examples/native_units /tmp/art 20 8 1024 2>&1 | grep -v amdgpu.ids | tail -2 )
done
