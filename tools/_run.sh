timeout 900 python -m pytest tests/test_net_gpu.py tests/test_backward_gpu.py tests/test_hostile_inputs_gpu.py tests/test_full_training_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -4
for v in 0 1; do echo "wgrad_winograd=$v"; python tools/train_bench.py --steps 6 --wgrad-winograd $v 2>&1 | tail -1 | cut -c1-140; done
python tools/step_ab.py --knob mpsr_debug_set_wino3_form --values 0,2 --rounds 3 2>&1 | tail -2
