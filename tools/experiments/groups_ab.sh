for G in 1 2 4; do for SS in "" "--side-stream"; do
IS_PW_GROUPS=$G python bench.py --preset drn_d_38_pairwise --no-variants --no-cpu-baseline --no-d2h --no-single --no-verify --no-prune-stats --min-seconds 1 $SS 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('groups $G $SS', round(d['value']), d['ms_per_step'])"
done; done
python bench.py --no-variants --no-cpu-baseline --no-d2h --no-single --no-verify --no-prune-stats --min-seconds 1 --side-stream 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unary side', round(d['value']), d['ms_per_step'])"
