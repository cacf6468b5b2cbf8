bash tools/r04_session3.sh; bash tools/r04_session4.sh
